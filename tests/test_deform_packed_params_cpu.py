"""scene.deformation._flat_stage: the packed parameter block the C ABI reads is kept current WITHOUT a copy per call (the
parameters' storage is the block itself); names / shapes / state dict as the reference's deform_network (scene/deformation.py:15-106)."""
import copy

import torch

from oracle import deformation_ref as R


def _net():
    from scene.deformation import deform_network
    a = R.Args(no_do=False, use_coarse_temporal_embedding=True, c2f_temporal_iter=10000, deform_from_iter=5000)
    torch.manual_seed(1)
    return deform_network(D=1, W=64, min_embeddings=30, max_embeddings=150, num_frames=300, args=a)


def test_packed_block_tracks_parameters_without_copies():
    net = _net()
    sd0 = copy.deepcopy(net.state_dict())
    f1 = net._flat_stage("c")
    assert f1.grad_fn is not None                      # autograd on: the view node
    store = net._flat_store["c"]
    assert net._flat_stage("c").data_ptr() == store.data_ptr() == f1.data_ptr()      # second call: same storage, no repack
    # values and order: feature_out.0.{weight,bias}, then per head {1.weight, 1.bias, 3.weight, 3.bias}
    parts = net._stage_parts("c")
    assert torch.equal(store, torch.cat([p.detach().reshape(-1) for p in parts]))
    for k, v in net.state_dict().items():
        assert torch.equal(v, sd0[k]), k                # packing changed no value, name or shape
    # an optimizer step (in place) is visible in the block
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    (f1 * torch.arange(f1.numel(), dtype=torch.float32)).sum().backward()
    w = net.pos_deform_c[3].weight
    assert w.grad is not None and w.grad.shape == w.shape
    off = sum(p.numel() for p in parts[:[id(p) for p in parts].index(id(w))])
    assert torch.equal(w.grad.reshape(-1), torch.arange(off, off + w.numel(), dtype=torch.float32))
    before = store.clone()
    opt.step()
    assert not torch.equal(store, before)
    assert torch.equal(net._flat_stage("c").detach(), torch.cat([p.detach().reshape(-1) for p in parts]))
    assert net._flat_store["c"].data_ptr() == store.data_ptr()
    # load_state_dict copies in place: still the same block
    net.load_state_dict(sd0)
    assert net._flat_store["c"].data_ptr() == store.data_ptr()
    assert torch.equal(net._flat_stage("c").detach(), torch.cat([sd0[k].reshape(-1) for k in
                                                                 ["feature_out_c.0.weight", "feature_out_c.0.bias"] +
                                                                 [f"{h}_deform_c.{i}.{wb}" for h in ("pos", "scales", "rotations", "opacity", "rgb")
                                                                  for i in (1, 3) for wb in ("weight", "bias")]]))


def test_rebound_storage_is_repacked():
    net = _net()
    net._flat_stage("f")
    old = net._flat_store["f"]
    net.double().float()                                 # every parameter gets fresh storage, as .to(device) does
    with torch.no_grad():
        net.opacity_deform_f[1].bias.add_(1.0)
    f = net._flat_stage("f")
    assert net._flat_store["f"].data_ptr() != old.data_ptr()
    assert torch.equal(f.detach(), torch.cat([p.detach().reshape(-1) for p in net._stage_parts("f")]))
    with torch.no_grad():
        assert net._flat_stage("f").grad_fn is None      # rendering: the block itself
