"""End-to-end CHAIN parity with the deformation ON (missing #4 / #5):

  * render()'s parameter gradients -- _xyz, _scaling, _rotation, _opacity, _features_dc, _features_rest, _embedding and every MLP
    parameter -- through raster backward -> activations backward -> deformation backward (split-SH and active-row paths
    composed), against the ORACLE CHAIN: the C oracle's backward fed the HIP forward's saved per-pixel state
    (oracle/raster_ref.c) -> torch activations -> oracle/deformation_torch.py autograd (the reference's chain:
    gaussian_renderer/__init__.py:74-109).  Tolerance 1e-4 of each tensor's largest element.
  * the C5 item (BASELINE.json configs[4]): 500k Gaussians, SH degree 3, 1080p, deformation on, all outputs (TTT) forward
    against the OpenMP oracle on the HIP deformation's own outputs: tile lists bit-exact, images 1e-4.
"""
import math

import numpy as np
import pytest
import torch

import util
from test_raster_parity_gpu import MARGIN, TOL_GRAD, TOL_IMG, _check_images, _check_state

pytestmark = pytest.mark.gpu
TOL_CHAIN = 1e-4


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.mark.parametrize("variant", ["FTT", "TTT"])
def test_render_parameter_gradients_vs_oracle_chain(variant):
    _need_gpu()
    from diff_gaussian_rasterization import _C
    from ed3dgs_amd import synthetic as S
    from ed3dgs_amd.model import PIPE, SynthGaussianModel, default_hyper
    from gaussian_renderer import render
    from oracle import deformation_torch as DT
    dev = "cuda"
    rc, rd = util.VARIANTS[variant]
    P, W, H = 5003, 256, 192
    t = 0.37
    scene = S.make_scene(P, seed=41)
    hy = default_hyper()
    model = SynthGaussianModel(scene, args=hy, device=dev)
    cam = S.make_cameras(2, W, H, seed=42, device=dev)[1].with_time(t)
    kw = dict(kernel_size=0.0, require_coord=rc, require_depth=rd, iter=20000, num_down_emb_c=30, num_down_emb_f=30)

    # ---- HIP: render() forward; keep the rasterizer's state to feed the oracle backward the same per-pixel state ----
    _C.KEEP_LAST = True
    try:
        pkg = render(cam, model, PIPE, torch.ones(3, device=dev), **kw)
        L = dict(_C.LAST)
        sv = _C.state_view(L["P"], L["H"], L["W"], L["R"], L["geom"], L["binning"], L["img"])
    finally:
        _C.KEEP_LAST = False
        _C.LAST.clear()

    # ---- oracle chain, forward.  The rasterizer oracle is fed the HIP deformation's OWN outputs (bit-identical inputs: the tile
    # lists of the two sides are then the bit-exact ones of tests/test_raster_parity_gpu.py); the deformation + activations graph
    # the gradients flow back through is the torch restatement (values within 1e-7 of the HIP ones) ----
    from ed3dgs_amd.activations import fused_activations
    with torch.no_grad():
        h_m3, h_sc, h_ro, h_op, h_sh, _ = model._deformation(model.get_xyz, model._scaling, model._rotation, model._opacity, t, None,
                                                            model, None, model.get_features, iter=20000, num_down_emb_c=30,
                                                            num_down_emb_f=30)
        h_sca, h_roa, h_opa = fused_activations(h_sc, h_ro, h_op, None)
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model._deformation.state_dict().items()}
    leaf = lambda x: x.detach().cpu().clone().requires_grad_(True)
    xyz, ls, rot, op, dc, rest, emb = [leaf(x) for x in (model._xyz, model._scaling, model._rotation, model._opacity,
                                                          model._features_dc, model._features_rest, model._embedding)]
    (m3, ls_f, rot_f, op_f, sh_f), _ = DT.forward(sd, hy, hy.defor_depth, hy.max_embeddings, xyz, ls, rot, op,
                                                  torch.cat((dc, rest), 1), emb, t, None, 20000, 30, 30)
    scales_a, rots_a, opac_a = torch.exp(ls_f), torch.nn.functional.normalize(rot_f), torch.sigmoid(op_f)
    for got, ref in ((h_m3, m3), (h_sca, scales_a), (h_roa, rots_a), (h_opa, opac_a), (h_sh, sh_f)):
        assert util.grad_err(got.cpu().numpy(), ref.detach().numpy().reshape(got.shape)) <= 1e-5
    inp = dict(P=P, W=W, H=H, bg=torch.ones(3), means3D=h_m3.cpu(), opacities=h_opa.cpu(), tongue_class=scene.tongue_class,
               scales=h_sca.cpu(), rotations=h_roa.cpu(), shs=h_sh.cpu(), viewmatrix=cam.world_view_transform.cpu(),
               projmatrix=cam.full_proj_transform.cpu(), campos=cam.camera_center.cpu(), tanfovx=math.tan(cam.FoVx * 0.5),
               tanfovy=math.tan(cam.FoVy * 0.5), kernel_size=0.0, scale_modifier=1.0, sh_degree=3)
    fw = util.oracle_forward(inp, variant)
    np.testing.assert_array_equal(sv["point_list"], fw["point_list"])
    np.testing.assert_array_equal(sv["ranges"], fw["ranges"])

    # ---- the same upstream gradients on both sides (pixels with a near-threshold blend decision get none) ----
    grads = util.zero_unused_grads(S.make_upstream_grads(H, W, seed=43), variant)
    grads, frac = util.mask_marginal(grads, fw, MARGIN)
    gd = {k: v.to(dev) for k, v in grads.items()}
    outs, ups = [pkg["render"], pkg["mask"]], [gd["color"], gd["alpha"]]
    if rd:
        outs += [pkg["expected_depth"], pkg["median_depth"]]; ups += [gd["depth"], gd["mdepth"]]
    if rc:
        outs += [pkg["expected_coord"], pkg["median_coord"]]; ups += [gd["coord"], gd["mcoord"]]
    if rc or rd:
        outs += [pkg["normal"]]; ups += [gd["normal"]]
    torch.autograd.backward(outs, ups)

    # ---- oracle chain, backward: C oracle backward on the HIP forward's saved state -> torch autograd down to the leaves ----
    hip_out = [None] * 13
    hip_out[4], hip_out[6] = pkg["mask"].detach(), pkg["normal"].detach()
    bw = util.oracle_backward(inp, util.oracle_state_from_hip(fw, hip_out, sv), grads, variant)
    tt = lambda a, like: torch.from_numpy(np.ascontiguousarray(a)).reshape(like.shape).to(like.dtype)
    torch.autograd.backward([m3, opac_a, scales_a, rots_a, sh_f],
                            [tt(bw["dL_dmeans3D"], m3), tt(bw["dL_dopacity"], opac_a), tt(bw["dL_dscales"], scales_a),
                             tt(bw["dL_drotations"], rots_a), tt(bw["dL_dsh"], sh_f)])

    errs = {}
    for name, got, ref in (("_xyz", model._xyz.grad, xyz.grad), ("_scaling", model._scaling.grad, ls.grad),
                           ("_rotation", model._rotation.grad, rot.grad), ("_opacity", model._opacity.grad, op.grad),
                           ("_features_dc", model._features_dc.grad, dc.grad), ("_features_rest", model._features_rest.grad, rest.grad),
                           ("_embedding", model._embedding.grad, emb.grad)):
        errs[name] = util.grad_err(got.cpu().numpy(), ref.numpy())
    for name, p in model._deformation.named_parameters():
        ref = sd[name].grad
        if ref is None or float(ref.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        errs["mlp." + name] = util.grad_err(p.grad.cpu().numpy(), ref.numpy())
    errs["viewspace_points"] = util.grad_err(pkg["viewspace_points"].grad.cpu().numpy(), bw["dL_dmeans2D"].reshape(P, 3))
    worst = max(errs, key=errs.get)
    print("chain", variant, "masked px frac %.2e" % frac, "worst", worst, errs[worst], {k: float("%.2e" % v) for k, v in errs.items()})
    for k, v in errs.items():
        assert v <= TOL_CHAIN, (k, v)


def test_c5_item_500k_all_outputs_forward_and_backward():
    """BASELINE.json configs[4], one item: 500k Gaussians, SH degree 3, 1080p, full deformation, depth / coord / normal outputs
    (TTT): deformation vs the golden-pinned numpy restatement, rasterizer on the HIP deformation's own outputs vs the OpenMP
    C oracle on those same numbers -- tile lists, ranges, keys bit-exact, images 1e-4 --, render() equal to the raw call, and the
    backward kernels (every plane's gradient in) against the oracle's backward at kernel level."""
    _need_gpu()
    from ed3dgs_amd import synthetic as S
    from ed3dgs_amd.activations import fused_activations
    from ed3dgs_amd.model import PIPE, SynthGaussianModel, default_hyper
    from gaussian_renderer import render
    from oracle import deformation_ref as R
    dev = "cuda"
    P, W, H, frames = 500_000, 1920, 1080, 150
    t = 71 / frames
    scene = S.make_scene(P, seed=0)
    hy = default_hyper(total_num_frames=frames)
    model = SynthGaussianModel(scene, args=hy, device=dev)
    cam = S.make_cameras(8, W, H, seed=1, device=dev)[3].with_time(t)
    with torch.no_grad():
        m3, sc_f, ro_f, op_f, sh_f, _ = model._deformation(model.get_xyz, model._scaling, model._rotation, model._opacity, t,
                                                          None, model, None, model.get_features, iter=20000,
                                                          num_down_emb_c=30, num_down_emb_f=30)
        sc_a, ro_a, op_a = fused_activations(sc_f, ro_f, op_f, None)
    sd = {k: v.detach().cpu().numpy() for k, v in model._deformation.state_dict().items()}
    fin, _, _ = R.forward(sd, hy, hy.defor_depth, hy.max_embeddings, scene.xyz.numpy(), scene.log_scale.numpy(),
                          scene.rot.numpy(), scene.opacity.numpy(), torch.cat((scene.f_dc, scene.f_rest), 1).numpy(),
                          scene.embedding.numpy(), t, None, 20000, 30, 30)
    for got, ref, name in ((m3, fin[0], "xyz"), (sc_f, fin[1], "scale"), (ro_f, fin[2], "rot"), (op_f, fin[3], "opacity"), (sh_f, fin[4], "sh")):
        err = np.abs(got.cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())
        assert err <= 1e-4, (name, err)
    inp = dict(P=P, W=W, H=H, bg=torch.ones(3), means3D=m3.cpu(), opacities=op_a.cpu(), tongue_class=scene.tongue_class,
               scales=sc_a.cpu(), rotations=ro_a.cpu(), shs=sh_f.cpu(), viewmatrix=cam.world_view_transform.cpu(),
               projmatrix=cam.full_proj_transform.cpu(), campos=cam.camera_center.cpu(), tanfovx=math.tan(cam.FoVx * 0.5),
               tanfovy=math.tan(cam.FoVy * 0.5), kernel_size=0.0, scale_modifier=1.0, sh_degree=3)
    fw = util.oracle_forward(inp, "TTT")
    out, sv = util.hip_forward_raw(inp, "TTT")
    _check_state(fw, out, sv)
    errs, frac = _check_images(fw, out, "TTT")
    good = fw["margin"] >= MARGIN
    np.testing.assert_array_equal(sv["n_contrib"][0][good], fw["n_contrib"][0][good])
    print("C5 item: R", fw["num_rendered"], "fwd rel-Linf", errs, "masked px frac %.2e" % frac)
    pkg = render(cam, model, PIPE, torch.ones(3, device=dev), kernel_size=0.0, require_coord=True, require_depth=True, iter=20000,
                 num_down_emb_c=30, num_down_emb_f=30)
    for k, i in (("render", 1), ("expected_coord", 2), ("median_coord", 3), ("mask", 4), ("normal", 6), ("expected_depth", 7),
                 ("median_depth", 8), ("radii", 9)):
        assert torch.equal(pkg[k].detach(), out[i]), k
    # backward at the same size, kernel level (K7 + K8 + K9 with every output plane's gradient): the C oracle's backward fed the
    # HIP forward's saved state, upstream gradients zeroed on the threshold-marginal pixels
    grads = util.zero_unused_grads(S.make_upstream_grads(H, W), "TTT")
    grads, _ = util.mask_marginal(grads, fw, MARGIN)
    bw = util.oracle_backward(inp, util.oracle_state_from_hip(fw, out, sv), grads, "TTT")
    got = util.hip_backward_raw(inp, out, grads, "TTT")
    gerr = {n: util.grad_err(got[n].reshape(bw[n].shape), bw[n]) for n in util.GRAD_NAMES}
    print("C5 item bwd (kernel level)", gerr)
    for n, v in gerr.items():
        assert v <= TOL_GRAD, (n, v)
