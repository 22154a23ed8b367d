"""GPU parity of the fused HIP deformation network (scene/deformation.py drop-in) against
  (1) the golden fixtures generated from the reference's own scene/deformation.py (values and autograd gradients),
  (2) the PyTorch-CPU restatement (oracle/deformation_torch.py, itself pinned by the same fixtures) at larger sizes.
Tolerance: 1e-4 relative L-inf (north_star), measured against each tensor's max magnitude."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "deform_*.npz")))
TOL = 1e-4
NAMES = ("xyz", "scales", "rot", "opacity", "sh")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


class _Args:
    pass


def _load(path):
    z = np.load(path)
    a = _Args()
    for k in z.files:
        if k.startswith("arg_"):
            setattr(a, k[4:], z[k].item())
    return z, a


def _build(z, a, device="cuda"):
    from scene.deformation import deform_network
    net = deform_network(D=int(z["cfg_D"]), W=int(z["cfg_W"]), min_embeddings=int(z["cfg_min"]),
                         max_embeddings=int(z["cfg_max"]), num_frames=300, args=a)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")}
    net.load_state_dict(sd, strict=True)  # state-dict compatibility with the reference (deformation.pth)
    return net.to(device)


class _PC:
    def __init__(self, e):
        self.get_embedding = e


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64).reshape(a.shape)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("mode", ["exact-split", "split-bf16x3", "fp32-mfma"])
@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[7:-4] for p in GOLD])
def test_golden_values_and_grads(path, mode, monkeypatch, libopt):
    """Against the reference's own outputs, tolerance 1e-4, in the three multiply modes of csrc/deform.hip: the default
    (three exact bf16 pieces per fp32 operand, eight products), the f32-operand MFMA kernels (ED3DGS_DEFORM_FP32_MFMA=1)
    and the opt-in reduced two-piece / three-product kernels (ED3DGS_DEFORM_BF16X3=1)."""
    _need_gpu()
    for v in ("DEFORM_BF16X3", "DEFORM_FP32_MFMA"):
        libopt(v, 0)
    if mode == "split-bf16x3":
        libopt("DEFORM_BF16X3", 1)
    elif mode == "fp32-mfma":
        libopt("DEFORM_FP32_MFMA", 1)
    z, a = _load(path)
    if int(z["cfg_D"]) > 1 and mode != "exact-split":
        pytest.skip("defor_depth > 1 takes the layer-by-layer fp32 path (csrc/deform_deep.hip) in every multiply mode")
    net = _build(z, a)
    leaf = lambda n: torch.from_numpy(z["in_" + n]).cuda().requires_grad_(True)
    xyz, scales, rot, opacity, sh, emb = [leaf(n) for n in ("xyz", "scales", "rot", "opacity", "sh", "emb")]
    cam = int(z["cfg_cam_no"]); cam = None if cam < 0 else cam
    outs = net(xyz, scales, rot, opacity, float(z["cfg_time"]), cam, _PC(emb), None, sh, iter=int(z["cfg_iter"]),
               num_down_emb_c=int(z["cfg_nde_c"]), num_down_emb_f=int(z["cfg_nde_f"]))
    final, sub = outs[:5], outs[5][0]
    errs = {}
    for n, t in zip(NAMES, final):
        errs["out_" + n] = rel(t.detach().cpu().numpy(), z["out_" + n])
    for n, t in zip(NAMES, sub):
        errs["sub_" + n] = rel(t.detach().cpu().numpy(), z["sub_" + n])
    ws = [torch.from_numpy(z[f"lossw_{i}"]).cuda() for i in range(10)]
    loss = sum((t * w.reshape(t.shape)).sum() for t, w in zip(list(final) + list(sub), ws))
    loss.backward()
    gerrs = {}
    for name, p in net.named_parameters():
        ref = z["gsd_" + name]
        if ref.size == 0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        got = torch.zeros_like(p) if p.grad is None else p.grad
        if np.abs(ref).max() == 0:
            assert float(got.abs().max()) < 1e-12, name
            continue
        gerrs[name] = rel(got.cpu().numpy(), ref)
    for n, t in zip(("xyz", "scales", "rot", "opacity", "sh", "emb"), (xyz, scales, rot, opacity, sh, emb)):
        gerrs["in_" + n] = rel(t.grad.cpu().numpy(), z["gin_" + n])
    print(os.path.basename(path), "value max", max(errs.values()), "grad max", max(gerrs.values()),
          "worst", max(gerrs, key=gerrs.get))
    for k, v in {**errs, **gerrs}.items():
        assert v <= TOL, (k, v)


# 65,836 = 512 * 128 + 300: more block iterations than the resident grid with a small remainder, i.e. the forward's
# (group, head) tail units run (deform_forward_pipe_kernel)
@pytest.mark.parametrize("P,W,keep,flags", [
    (5000, 128, True, {}), (5000, 128, False, {}), (777, 64, True, {}), (65836, 128, True, {}),
    # tail units with disabled heads (their tensors pass through with head 0's unit) and with one stage only
    (65836, 128, True, dict(no_dr=True, no_dc=True)), (65836, 128, True, dict(no_coarse_deform=True, no_ds=True)),
    (66100, 128, False, dict(no_fine_deform=True)),
    # the opt-in split-bf16 MFMA kernels (forward, kept data gradient, narrow-head weight gradients)
    (5000, 128, True, dict(b3=True)), (65836, 128, True, dict(b3=True, no_dr=True)), (777, 64, True, dict(b3=True)),
    # the f32-operand MFMA kernels
    (5000, 128, True, dict(f32=True)), (5000, 128, False, dict(f32=True)), (65836, 128, True, dict(f32=True, no_dr=True)),
    (777, 64, True, dict(f32=True)),
    # the backward over the active rows only (csrc/deform.hip deform_active_rows_body): most upstream rows exactly zero, as
    # render()'s backward delivers them for occluded / culled Gaussians; a single active row; none at all; and the dense
    # walk (ED3DGS_DEFORM_DENSE_BWD=1) of the same sparse case
    (65836, 128, True, dict(sparse=0.6)), (70001, 128, True, dict(sparse=0.97, no_dr=True)), (5000, 128, True, dict(sparse=-1)),
    (5000, 128, True, dict(sparse=1.0)), (65836, 128, True, dict(sparse=0.6, dense_bwd=True)),
    (65836, 128, True, dict(sparse=0.6, no_coarse_deform=True)),
    # the head weight gradients + the per-frame backward as separate launches (ED3DGS_WGRAD_SEPARATE=1; default: one launch,
    # which every case above runs)
    (65836, 128, True, dict(separate_wgrad=True)), (5000, 128, True, dict(sparse=0.6, separate_wgrad=True)),
], ids=["5k", "5k-stateless", "w64", "tail", "tail-no_dr-no_dc", "tail-fine-only-no_ds", "tail-coarse-only-stateless",
        "5k-split-bf16", "tail-split-bf16-no_dr", "w64-split-bf16",
        "5k-fp32-mfma", "5k-stateless-fp32-mfma", "tail-fp32-mfma-no_dr", "w64-fp32-mfma",
        "sparse60", "sparse97-no_dr", "one-active-row", "no-active-row", "sparse60-dense-walk", "sparse60-fine-only",
        "tail-wgrad-launches-separate", "5k-sparse60-wgrad-launches-separate"])
def test_against_torch_restatement(P, W, keep, flags, monkeypatch, libopt):
    """keep=True: the forward keeps the hidden activations for the backward (width 128; other widths re-form them);
    keep=False: the stateless backward that re-forms them.  Both against the float64 restatement."""
    _need_gpu()
    from oracle import deformation_ref as R
    from oracle import deformation_torch as T
    import scene.deformation as SD
    from scene.deformation import deform_network
    monkeypatch.setattr(SD, "KEEP_ACTIVATIONS", keep)
    flags = dict(flags)
    if flags.pop("b3", False):
        libopt("DEFORM_BF16X3", 1)
    if flags.pop("f32", False):
        libopt("DEFORM_FP32_MFMA", 1)
    sparse = flags.pop("sparse", 0.0)
    libopt("WGRAD_SEPARATE", 1 if flags.pop("separate_wgrad", False) else 0)
    libopt("DEFORM_DENSE_BWD", 0)
    if flags.pop("dense_bwd", False):
        libopt("DEFORM_DENSE_BWD", 1)
    a = R.Args(**{**dict(no_do=False, use_coarse_temporal_embedding=True, c2f_temporal_iter=10000, deform_from_iter=5000), **flags})
    torch.manual_seed(5)
    net = deform_network(D=1, W=W, min_embeddings=30, max_embeddings=150, num_frames=300, args=a)
    with torch.no_grad():
        net.weight.mul_(100.0)
    g = torch.Generator().manual_seed(6)
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    base = dict(xyz=mk(P, 3), scales=mk(P, 3, sc=0.3) - 4, rot=mk(P, 4), opacity=mk(P, 1), sh=mk(P, 16, 3, sc=0.5),
                emb=mk(P, 32, sc=0.1))
    it, t = 20000, 0.43
    # CPU restatement (float64) with autograd
    sd64 = {k: v.detach().double().requires_grad_(True) for k, v in net.state_dict().items()}
    b64 = {k: v.double().requires_grad_(True) for k, v in base.items()}
    mg = []
    fin, sub = T.forward(sd64, a, 1, 150, b64["xyz"], b64["scales"], b64["rot"], b64["opacity"], b64["sh"], b64["emb"],
                         t, None, it, 30, 30, margin_out=mg)
    # Gaussians sitting on a ReLU kink (|pre-activation| within fp32 rounding of 0) have a discontinuous gradient:
    # they are taken out of the loss (both paths), so every compared gradient is well-conditioned.
    off_kink = (mg[0] > 1e-6).float()
    assert off_kink.mean() > 0.9
    if sparse > 0:       # rows whose upstream gradient is exactly zero in every tensor
        off_kink = off_kink * (torch.rand(P, generator=g) >= sparse).float()
    elif sparse < 0:     # exactly one active row
        one = torch.zeros(P); one[int(off_kink.argmax()) if P < 3000 else int(torch.nonzero(off_kink)[2999])] = 1.0
        off_kink = off_kink * one
    ws = [mk(*x.shape) * off_kink.reshape(-1, *([1] * (x.dim() - 1))) for x in list(fin) + list(sub)]
    loss = sum((x * w.double()).sum() for x, w in zip(list(fin) + list(sub), ws))
    loss.backward()
    # HIP
    net = net.cuda()
    bg = {k: v.cuda().requires_grad_(True) for k, v in base.items()}
    outs = net(bg["xyz"], bg["scales"], bg["rot"], bg["opacity"], t, None, _PC(bg["emb"]), None, bg["sh"], iter=it,
               num_down_emb_c=30, num_down_emb_f=30)
    hf, hs = outs[:5], outs[5][0]
    lossg = sum((x * w.cuda().reshape(x.shape)).sum() for x, w in zip(list(hf) + list(hs), ws))
    lossg.backward()
    errs = {}
    for n, x, y in zip(NAMES, hf, fin):
        errs["out_" + n] = rel(x.detach().cpu().numpy(), y.detach().numpy())
    for n, x, y in zip(NAMES, hs, sub):
        errs["sub_" + n] = rel(x.detach().cpu().numpy(), y.detach().numpy())
    for name, p in net.named_parameters():
        ref = sd64[name].grad
        if ref is None or float(ref.abs().max()) == 0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        errs["g_" + name] = rel(p.grad.cpu().numpy(), ref.numpy())
    for k in base:
        if float(b64[k].grad.abs().max()) == 0:
            assert float(bg[k].grad.abs().max()) == 0.0, k
            continue
        errs["gin_" + k] = rel(bg[k].grad.cpu().numpy(), b64[k].grad.numpy())
    print("P", P, "W", W, "max err", max(errs.values()), "worst", max(errs, key=errs.get))
    for k, v in errs.items():
        assert v <= TOL, (k, v)


@pytest.mark.parametrize("P,flags,loss_on_sub", [(5000, {}, True), (65836, {}, False), (5000, dict(no_dc=True), True),
                                                 (5000, dict(f32=True), True), (777, dict(stateless=True), True)],
                         ids=["5k", "tail-no-sub-loss", "no_dc", "fp32-mfma", "stateless"])
def test_split_sh_storage_matches_whole(P, flags, loss_on_sub, monkeypatch, libopt):
    """forward(sh_coefs=_features_dc, sh_coefs_rest=_features_rest) against forward(sh_coefs=cat(dc, rest)): identical
    outputs (bit for bit: the same arithmetic on the same values), dL/d dc and dL/d rest equal to the slices of the whole
    tensor's gradient (bit for bit: g_sh + gs_sh either way), parameter gradients equal up to the order of the atomic sums."""
    _need_gpu()
    from oracle import deformation_ref as R
    import scene.deformation as SD
    from scene.deformation import deform_network
    flags = dict(flags)
    if flags.pop("f32", False):
        libopt("DEFORM_FP32_MFMA", 1)
    monkeypatch.setattr(SD, "KEEP_ACTIVATIONS", not flags.pop("stateless", False))
    a = R.Args(**{**dict(no_do=False, use_coarse_temporal_embedding=True, c2f_temporal_iter=10000, deform_from_iter=5000), **flags})
    torch.manual_seed(11)
    net = deform_network(D=1, W=128, min_embeddings=30, max_embeddings=150, num_frames=300, args=a).cuda()
    with torch.no_grad():
        net.weight.mul_(100.0)
    g = torch.Generator().manual_seed(12)
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).cuda()
    base = dict(xyz=mk(P, 3), scales=mk(P, 3, sc=0.3) - 4, rot=mk(P, 4), opacity=mk(P, 1), dc=mk(P, 1, 3), rest=mk(P, 15, 3, sc=0.2),
                emb=mk(P, 32, sc=0.1))
    ws = [mk(P, 3), mk(P, 3), mk(P, 4), mk(P, 1), mk(P, 16, 3)]
    ws_sub = [mk(P, 3), mk(P, 3), mk(P, 4), mk(P, 1), mk(P, 16, 3)]
    dead = (torch.rand(P, generator=g) < 0.5).cuda()          # rows without any upstream gradient, as render() delivers them
    for w in ws + ws_sub:
        w[dead] = 0

    def run(split):
        net.zero_grad(set_to_none=True)
        b = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        if split:
            outs = net(b["xyz"], b["scales"], b["rot"], b["opacity"], 0.37, None, _PC(b["emb"]), None, b["dc"], iter=20000,
                       num_down_emb_c=30, num_down_emb_f=30, sh_coefs_rest=b["rest"])
            assert isinstance(outs[5][1][4], tuple) and outs[5][1][4][1] is b["rest"]
        else:
            whole = torch.cat((b["dc"], b["rest"]), 1)
            outs = net(b["xyz"], b["scales"], b["rot"], b["opacity"], 0.37, None, _PC(b["emb"]), None, whole, iter=20000,
                       num_down_emb_c=30, num_down_emb_f=30)
        fin, sub = outs[:5], outs[5][0]
        loss = sum((x * w).sum() for x, w in zip(fin, ws))
        if loss_on_sub:
            loss = loss + sum((x * w).sum() for x, w in zip(sub, ws_sub))
        loss.backward()
        return ([x.detach().clone() for x in list(fin) + list(sub)], {k: v.grad.clone() for k, v in b.items()},
                {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})

    o1, g1, p1 = run(False)
    o2, g2, p2 = run(True)
    for x, y in zip(o1, o2):
        assert torch.equal(x, y)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
    assert float(g2["dc"][dead].abs().max()) == 0.0 and float(g2["rest"][~dead].abs().max()) > 0
    for n in p1:
        assert float((p1[n] - p2[n]).abs().max()) <= 1e-5 * max(float(p1[n].abs().max()), 1e-30), n


@pytest.mark.parametrize("release", [False, True], ids=["workspace-kept", "workspace-released"])
def test_second_backward_over_a_retained_graph(release, monkeypatch):
    """retain_graph=True and a second backward.  Workspace kept with the graph: the active-row counters in it were re-armed
    by the first backward's last block, so the second walk sees the same rows -- dL/d inputs exactly double (they are
    deterministic), the weight gradients double up to the order of their atomic sums.  Workspace released by the first
    backward (the default): the second one re-forms the activations (stateless kernels) -- double within the tolerance."""
    _need_gpu()
    from oracle import deformation_ref as R
    import scene.deformation as SD
    from scene.deformation import deform_network
    monkeypatch.setattr(SD, "RELEASE_KEPT_WORKSPACE", release)
    a = R.Args(no_do=False, use_coarse_temporal_embedding=True, c2f_temporal_iter=10000, deform_from_iter=5000)
    torch.manual_seed(21)
    net = deform_network(D=1, W=128, min_embeddings=30, max_embeddings=150, num_frames=300, args=a).cuda()
    g = torch.Generator().manual_seed(22)
    P = 9001
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).cuda().requires_grad_(True)
    xyz, sc_, rot, op, sh, emb = mk(P, 3), mk(P, 3, sc=0.3), mk(P, 4), mk(P, 1), mk(P, 16, 3, sc=0.5), mk(P, 32, sc=0.1)
    outs = net(xyz, sc_, rot, op, 0.61, None, _PC(emb), None, sh, iter=20000, num_down_emb_c=30, num_down_emb_f=30)
    w = (torch.rand(P, generator=g) < 0.4).float().cuda()
    loss = sum((x.reshape(P, -1) * w[:, None]).sum() for x in outs[:5])
    loss.backward(retain_graph=True)
    e1 = emb.grad.clone()
    p1 = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    assert float(e1[w == 0].abs().max()) == 0.0 and float(e1[w > 0].abs().max()) > 0
    loss.backward()
    if release:
        assert float((emb.grad - 2 * e1).abs().max()) <= TOL * float(e1.abs().max())
    else:
        assert torch.equal(emb.grad, 2 * e1)
    for n, p in net.named_parameters():
        if n in p1 and float(p1[n].abs().max()) > 0:
            assert float((p.grad - 2 * p1[n]).abs().max()) <= (TOL if release else 1e-5) * float(p1[n].abs().max()), n


@pytest.mark.parametrize("with_filter", [False, True])
def test_fused_activations_match_torch(with_filter):
    """a7 (scene/gaussian_model.py:37-45, 594-603): the fused HIP activation kernel against the torch ops the
    reference uses for them (values and gradients), fp32, tolerance 1e-5 relative."""
    _need_gpu()
    from ed3dgs_amd.activations import fused_activations
    g = torch.Generator().manual_seed(3)
    P = 5003
    s = (torch.randn(P, 3, generator=g) * 0.5 - 3).cuda().requires_grad_(True)
    r = torch.randn(P, 4, generator=g).cuda().requires_grad_(True)
    o = (torch.randn(P, 1, generator=g) * 2).cuda().requires_grad_(True)
    f = (torch.rand(P, 1, generator=g) * 0.05).cuda() if with_filter else None
    ws = [torch.randn(P, 3, generator=g).cuda(), torch.randn(P, 4, generator=g).cuda(), torch.randn(P, 1, generator=g).cuda()]
    so, ro, oo = fused_activations(s, r, o, f)
    (so * ws[0]).sum().add((ro * ws[1]).sum()).add((oo * ws[2]).sum()).backward()
    got = [so.detach(), ro.detach(), oo.detach(), s.grad.clone(), r.grad.clone(), o.grad.clone()]
    s2, r2, o2 = [t.detach().clone().requires_grad_(True) for t in (s, r, o)]
    rr = torch.nn.functional.normalize(r2)
    if with_filter:
        sc = torch.exp(s2); sq = torch.square(sc); aq = sq + torch.square(f)
        ss = torch.sqrt(aq); oo2 = torch.sigmoid(o2) * torch.sqrt(sq.prod(dim=1) / aq.prod(dim=1))[..., None]
    else:
        ss = torch.exp(s2); oo2 = torch.sigmoid(o2)
    (ss * ws[0]).sum().add((rr * ws[1]).sum()).add((oo2 * ws[2]).sum()).backward()
    ref = [ss.detach(), rr.detach(), oo2.detach(), s2.grad, r2.grad, o2.grad]
    for a, b in zip(got, ref):
        assert float((a - b).abs().max()) <= 1e-5 * max(float(b.abs().max()), 1e-30)


@pytest.mark.parametrize("D", [1, 2, 3])
def test_train_style_steps_match_the_unpacked_module(D):
    """ADVICE r2 (_flat_stage re-binds every Linear parameter to a slice of one packed buffer): three full train-style steps
    -- forward, backward, Adam step, zero_grad(set_to_none=False) -- then a state_dict save / load into a fresh module, against
    the same steps on the float64 CPU restatement with ordinary, unpacked parameters (oracle/deformation_torch.py).
    defor_depth 1 (fused kernels) and 2, 3 (csrc/deform_deep.hip)."""
    _need_gpu()
    import io
    from oracle import deformation_ref as R
    from oracle import deformation_torch as T
    from scene.deformation import deform_network
    a = R.Args(no_do=False, use_coarse_temporal_embedding=True, c2f_temporal_iter=10000, deform_from_iter=5000)
    torch.manual_seed(31)
    net = deform_network(D=D, W=128 if D == 1 else 64, min_embeddings=30, max_embeddings=150, num_frames=300, args=a)
    with torch.no_grad():
        net.weight.mul_(100.0)
    ref_sd = {k: v.detach().double().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    net = net.cuda()
    g = torch.Generator().manual_seed(32)
    P = 3001
    mk = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    base = [mk(P, 3), mk(P, 3, sc=0.3) - 4, mk(P, 4), mk(P, 1), mk(P, 16, 3, sc=0.5)]
    emb = mk(P, 32, sc=0.1)
    ws = [mk(*t.shape) for t in base]
    opt_g = torch.optim.Adam(net.parameters(), lr=1e-3)   # (stale packed parameters would show in the next steps' gradients)
    opt_r = torch.optim.Adam(list(ref_sd.values()), lr=1e-3)
    names = [n for n, _ in net.named_parameters()]
    for step in range(3):
        t = 0.2 + 0.3 * step
        mg = []
        fin, _ = T.forward(ref_sd, a, D, 150, *[b.double() for b in base], emb.double(), t, None, 20000, 30, 30, margin_out=mg)
        # Gaussians on a ReLU kink (a pre-activation within fp32 rounding of zero) have a discontinuous gradient: out of the loss
        # on both sides, as in test_against_torch_restatement (one flipped unit of one Gaussian is 1 / P of a frame-level gradient)
        off_kink = (mg[0] > 1e-6).float()
        assert off_kink.mean() > 0.9
        wk = [w * off_kink.reshape(-1, *([1] * (w.dim() - 1))) for w in ws]
        sum((o.reshape(w.shape) * w.double()).sum() for o, w in zip(fin, wk)).backward()
        outs = net(*[b.cuda() for b in base[:4]], t, None, _PC(emb.cuda()), None, base[4].cuda(), iter=20000, num_down_emb_c=30, num_down_emb_f=30)
        sum((o * w.cuda()).sum() for o, w in zip(outs[:5], wk)).backward()
        for n in names:
            gr, gg = ref_sd[n].grad, dict(net.named_parameters())[n].grad
            if gr is None or float(gr.abs().max()) == 0:
                continue
            assert rel(gg.cpu().numpy(), gr.numpy()) <= TOL, (step, n, rel(gg.cpu().numpy(), gr.numpy()))
        opt_g.step(); opt_r.step()
        opt_g.zero_grad(set_to_none=False); opt_r.zero_grad(set_to_none=False)
        for n, p in net.named_parameters():
            assert p.dtype == torch.float32 and float(p.grad.abs().max()) == 0.0
    # Adam's first steps are +-lr whatever the gradient's size, so an element whose gradient is rounding noise around zero may
    # step the other way: all elements within 3 steps' reach, all but a sliver equal to 1e-5
    for n, p in net.named_parameters():
        diff = (p.detach().cpu().double() - ref_sd[n].detach()).abs()
        assert float(diff.max()) <= 3 * 2e-3 + 1e-6, n
        assert float((diff > 1e-5).float().mean()) <= 0.02, (n, float((diff > 1e-5).float().mean()))
    buf = io.BytesIO()
    torch.save(net.state_dict(), buf)
    buf.seek(0)
    net2 = deform_network(D=D, W=128 if D == 1 else 64, min_embeddings=30, max_embeddings=150, num_frames=300, args=a)
    net2.load_state_dict(torch.load(buf, weights_only=True))
    net2 = net2.cuda()
    with torch.no_grad():
        o1 = net(*[b.cuda() for b in base[:4]], 0.5, None, _PC(emb.cuda()), None, base[4].cuda(), iter=20000)
        o2 = net2(*[b.cuda() for b in base[:4]], 0.5, None, _PC(emb.cuda()), None, base[4].cuda(), iter=20000)
    for x, y in zip(o1[:5], o2[:5]):
        assert torch.equal(x, y)


def test_non_fp32_parameters_are_refused():
    _need_gpu()
    from oracle import deformation_ref as R
    from scene.deformation import deform_network
    a = R.Args(no_do=False)
    net = deform_network(D=1, W=64, min_embeddings=30, max_embeddings=150, num_frames=300, args=a).cuda().half()
    with pytest.raises(TypeError, match="fp32"):
        net(torch.zeros(4, 3).cuda(), torch.zeros(4, 3).cuda(), torch.zeros(4, 4).cuda(), torch.zeros(4, 1).cuda(),
            0.5, None, _PC(torch.zeros(4, 32).cuda()), None, torch.zeros(4, 16, 3).cuda(), iter=10)


@pytest.mark.parametrize("P,D,flags,with_filter", [
    (5000, 1, {}, False), (65836, 1, {}, False), (5000, 1, {}, True), (5000, 1, dict(f32=True), False), (5000, 1, dict(dense=True), False),
    (777, 1, dict(stateless=True), False), (3000, 2, {}, False), (5000, 1, dict(no_ds=True, no_dr=True), False),
    (5000, 1, dict(no_fine_deform=True), False)],
    ids=["5k", "tail-units", "filter3d", "fp32-mfma", "dense-walk", "stateless", "depth2", "no_ds-no_dr", "coarse-only"])
def test_activations_inside_the_deformation_match_the_stand_alone_launch(P, D, flags, with_filter, monkeypatch, libopt):
    """forward(..., activated=(filter,)) -- the MLP kernel's epilogue writes exp / normalize / sigmoid of the final values and the
    activation backward runs inside the deformation backward's first pass (ed3dgs_deform_forward_activated / _backward_activated)
    -- against forward(...) followed by the fused activation launch (ed3dgs_amd.activations, itself held to torch in
    test_fused_activations_match_torch): same values bit for bit (one activation_math.h), base-tensor / embedding gradients equal,
    parameter gradients equal up to the order of the atomic sums.  Covers the tail units (P = 65 836), the 3D-filter variant and
    the configurations that take the stand-alone launch inside the library (fp32-MFMA kernels: in-kernel too; dense walk,
    stateless, defor_depth 2: behind the network)."""
    _need_gpu()
    from oracle import deformation_ref as R
    import scene.deformation as SD
    from ed3dgs_amd.activations import fused_activations
    from scene.deformation import deform_network
    flags = dict(flags)
    if flags.pop("f32", False):
        libopt("DEFORM_FP32_MFMA", 1)
    if flags.pop("dense", False):
        libopt("DEFORM_DENSE_BWD", 1)
    monkeypatch.setattr(SD, "KEEP_ACTIVATIONS", not flags.pop("stateless", False))
    a = R.Args(**{**dict(no_do=False, use_coarse_temporal_embedding=True, c2f_temporal_iter=10000, deform_from_iter=5000), **flags})
    torch.manual_seed(21)
    W = 128 if D == 1 else 64
    net = deform_network(D=D, W=W, min_embeddings=30, max_embeddings=150, num_frames=300, args=a).cuda()
    with torch.no_grad():
        net.weight.mul_(100.0)
    g = torch.Generator().manual_seed(22)
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).cuda()
    base = dict(xyz=mk(P, 3), scales=mk(P, 3, sc=0.3) - 4, rot=mk(P, 4), opacity=mk(P, 1), dc=mk(P, 1, 3), rest=mk(P, 15, 3, sc=0.2),
                emb=mk(P, 32, sc=0.1))
    filt = (torch.rand(P, 1, generator=g) * 0.02).cuda() if with_filter else None
    ws = [mk(P, 3), mk(P, 3), mk(P, 4), mk(P, 1), mk(P, 16, 3)]
    ws_sub = [mk(P, 3), mk(P, 3), mk(P, 4), mk(P, 1), mk(P, 16, 3)]
    dead = (torch.rand(P, generator=g) < 0.5).cuda()
    for w in ws + ws_sub:
        w[dead] = 0

    def run(inside):
        net.zero_grad(set_to_none=True)
        b = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        kw = dict(activated=(filt,)) if inside else {}
        outs = net(b["xyz"], b["scales"], b["rot"], b["opacity"], 0.37, None, _PC(b["emb"]), None, b["dc"], iter=20000,
                   num_down_emb_c=30, num_down_emb_f=30, sh_coefs_rest=b["rest"], **kw)
        fin, sub = list(outs[:5]), outs[5][0]
        if not inside:
            fin[1], fin[2], fin[3] = fused_activations(fin[1], fin[2], fin[3], filt)
        loss = sum((x * w).sum() for x, w in zip(fin, ws)) + sum((x * w).sum() for x, w in zip(sub, ws_sub))
        loss.backward()
        return ([x.detach().clone() for x in fin + list(sub)], {k: v.grad.clone() for k, v in b.items()},
                {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})

    o1, g1, p1 = run(False)
    o2, g2, p2 = run(True)
    for i, (x, y) in enumerate(zip(o1, o2)):
        assert torch.equal(x, y), i
    assert float((o2[2].norm(dim=1) - 1).abs().max()) < 1e-5 and float(o2[1].min()) > 0 and 0 < float(o2[3].min()) and float(o2[3].max()) < 1
    for k in g1:
        assert float((g1[k] - g2[k]).abs().max()) <= 1e-6 * max(float(g1[k].abs().max()), 1e-30), k
    assert float(g2["scales"][dead].abs().max()) == 0.0 and float(g2["scales"][~dead].abs().max()) > 0
    for n in p1:
        assert float((p1[n] - p2[n]).abs().max()) <= 1e-5 * max(float(p1[n].abs().max()), 1e-30), n
