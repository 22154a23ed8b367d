"""GPU tests of the render() variants the first round left without one:
  * render_tongue / render_without_tongue (gaussian_renderer/__init__.py:145-287, 289-431): the masked subset through
    the whole path vs the oracle on that subset -- images, subset-length radii, viewspace_points.grad (full-P tensor,
    non-zero only on the masked rows, equal to the oracle's dL_dmeans2D incl. the abs-grad .z column);
  * pipe.debug = True (the CHECK_CUDA stage checks, CR/auxiliary.h:404-411): forward + backward, same bits as debug off;
  * odd Gaussian count with a poisoned allocator (api.hip gradient-record memset);
  * the C4 item (BASELINE.json configs[3]): 200k Gaussians, NeRSemble-shaped 1100x1604, deformation on, FTT.
"""
import math

import numpy as np
import pytest
import torch

import util
from test_raster_parity_gpu import MARGIN, TOL_GRAD, TOL_IMG, _check_images, _check_state

pytestmark = pytest.mark.gpu
MAX_MASKED_FRAC = 3e-4


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _static_model(P, seed, device="cuda"):
    """Deformation bypassed (no_coarse_deform + no_fine_deform: query_time returns its inputs, scene/deformation.py:125-133),
    so the rasterizer's inputs are the activated scene and the oracle can be fed the same numbers."""
    from ed3dgs_amd import synthetic as S
    from ed3dgs_amd.model import SynthGaussianModel, default_hyper
    sc = S.make_scene(P, seed=seed)
    g = torch.Generator().manual_seed(7)
    sc.tongue_class = (torch.rand(P, 1, generator=g) > 0.6).float()
    return sc, SynthGaussianModel(sc, args=default_hyper(no_coarse_deform=True, no_fine_deform=True), device=device)


@pytest.mark.parametrize("which", ["tongue", "without_tongue"])
def test_masked_render_variants_vs_oracle(which):
    _need_gpu()
    from diff_gaussian_rasterization import _C
    from ed3dgs_amd import synthetic as S
    from ed3dgs_amd.model import PIPE
    import gaussian_renderer as GR
    P, W, H = 6001, 320, 240
    sc, model = _static_model(P, seed=21)
    cam = S.make_cameras(3, W, H, seed=5, device="cuda")[1].with_time(0.25)
    fn = GR.render_tongue if which == "tongue" else GR.render_without_tongue
    is_tongue = torch.round(sc.tongue_class).bool().reshape(-1)
    mask = is_tongue if which == "tongue" else ~is_tongue
    n_sub = int(mask.sum())
    assert 0 < n_sub < P

    _C.KEEP_LAST = True
    try:
        pkg = fn(cam, model, PIPE, torch.ones(3, device="cuda"), kernel_size=0.1, require_coord=True,
                 require_depth=True, iter=20000, num_down_emb_c=30, num_down_emb_f=30)
        L = dict(_C.LAST)
        sv = _C.state_view(L["P"], L["H"], L["W"], L["R"], L["geom"], L["binning"], L["img"])
    finally:
        _C.KEEP_LAST = False
        _C.LAST.clear()
    assert L["P"] == n_sub                                     # the rasterizer saw the subset only
    assert pkg["radii"].shape == (n_sub,) and pkg["visibility_filter"].shape == (n_sub,)
    assert pkg["viewspace_points"].shape == (P, 3)             # ... while train.py reads a full-P gradient

    a = S.activated(sc)
    inp = dict(P=n_sub, W=W, H=H, bg=torch.ones(3), means3D=sc.xyz[mask], opacities=a["opacities"][mask],
               tongue_class=sc.tongue_class[mask], scales=a["scales"][mask], rotations=a["rotations"][mask],
               shs=a["shs"][mask], viewmatrix=cam.world_view_transform.cpu(), projmatrix=cam.full_proj_transform.cpu(),
               campos=cam.camera_center.cpu(), tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
               kernel_size=0.1, scale_modifier=1.0, sh_degree=3)
    fw = util.oracle_forward(inp, "TTT")
    np.testing.assert_array_equal(pkg["radii"].cpu().numpy(), fw["radii"])
    np.testing.assert_array_equal(sv["point_list"], fw["point_list"])
    np.testing.assert_array_equal(sv["ranges"], fw["ranges"])
    out = (L["R"], pkg["render"], pkg["expected_coord"], pkg["median_coord"], pkg["mask"], pkg["tongue_mask"],
           pkg["normal"], pkg["expected_depth"], pkg["median_depth"])
    errs, frac = _check_images(fw, [o.detach() if torch.is_tensor(o) else o for o in out], "TTT")
    if which == "tongue":                                      # every rendered Gaussian has class 1: tongue plane == alpha
        assert torch.allclose(pkg["tongue_mask"], pkg["mask"], atol=1e-6)
    else:
        assert float(pkg["tongue_mask"].abs().max()) == 0.0
    print(which, "subset", n_sub, "of", P, "fwd rel-Linf", errs, "masked px frac %.2e" % frac)
    assert frac <= MAX_MASKED_FRAC

    # backward through render(): the oracle backward is fed the HIP forward's saved state (kernel-level comparison)
    grads = util.zero_unused_grads(S.make_upstream_grads(H, W, seed=8), "TTT")
    grads, _ = util.mask_marginal(grads, fw, MARGIN)
    outs = [pkg["render"], pkg["expected_coord"], pkg["median_coord"], pkg["expected_depth"], pkg["median_depth"],
            pkg["mask"], pkg["normal"]]
    ups = [grads[k].cuda() for k in ("color", "coord", "mcoord", "depth", "mdepth", "alpha", "normal")]
    torch.autograd.backward(outs, ups)
    vg = pkg["viewspace_points"].grad
    assert vg is not None and vg.shape == (P, 3)
    assert float(vg[~mask.cuda()].abs().max()) == 0.0          # nothing outside the subset
    fake_out = [None] * 13
    fake_out[4], fake_out[6] = pkg["mask"].detach(), pkg["normal"].detach()
    bw = util.oracle_backward(inp, util.oracle_state_from_hip(fw, fake_out, sv), grads, "TTT")
    e = util.grad_err(vg[mask.cuda()].cpu().numpy(), bw["dL_dmeans2D"])
    assert (vg[:, 2] >= 0).all() and float(vg[:, 2].max()) > 0  # Q9: |dx|+|dy| accumulates in .z
    assert e <= TOL_GRAD, ("viewspace_points.grad", e)
    # the raw parameters' gradients: xyz through the (bypassed) deformation is the rasterizer's dL_dmeans3D scattered
    # back to the full-P tensor; rows outside the subset receive exactly zero
    gx = model._xyz.grad
    assert float(gx[~mask.cuda()].abs().max()) == 0.0
    e3 = util.grad_err(gx[mask.cuda()].cpu().numpy(), bw["dL_dmeans3D"])
    assert e3 <= TOL_GRAD, ("dL_dmeans3D", e3)
    gsh = torch.cat((model._features_dc.grad, model._features_rest.grad), 1)
    esh = util.grad_err(gsh[mask.cuda()].cpu().numpy(), bw["dL_dsh"])
    assert esh <= TOL_GRAD, ("dL_dsh", esh)
    print(which, "bwd rel-Linf means2D %.2e means3D %.2e sh %.2e" % (e, e3, esh))


def test_debug_true_forward_backward():
    """pipe.debug -> GaussianRasterizationSettings.debug -> a stream synchronise + error check after every stage
    (csrc/api.hip StageCheck; CR/auxiliary.h:404-411).  Same kernels, so the forward must be bit-identical to
    debug = False and the gradients equal up to the atomics' summation order."""
    _need_gpu()
    from types import SimpleNamespace
    from ed3dgs_amd import synthetic as S
    from ed3dgs_amd.model import PIPE, SynthGaussianModel
    from gaussian_renderer import render
    sc = S.make_scene(5000, seed=12)
    model = SynthGaussianModel(sc, device="cuda")
    cam = S.make_cameras(2, 333, 222, seed=2, device="cuda")[1].with_time(0.6)
    pipe_dbg = SimpleNamespace(**{**vars(PIPE), "debug": True})
    kw = dict(kernel_size=0.0, require_coord=True, require_depth=True, iter=20000, num_down_emb_c=30, num_down_emb_f=30)
    res = {}
    for name, pipe in (("off", PIPE), ("on", pipe_dbg)):
        for p in model.parameters():
            p.grad = None
        pkg = render(cam, model, pipe, torch.ones(3, device="cuda"), **kw)
        (pkg["render"].mean() + 0.1 * pkg["expected_depth"].mean() + 0.01 * pkg["normal"].sum() + pkg["expected_coord"].mean()).backward()
        res[name] = (pkg, [p.grad.clone() for p in model.parameters()], pkg["viewspace_points"].grad.clone())
    for k in ("render", "mask", "expected_depth", "median_depth", "normal", "expected_coord", "median_coord", "radii"):
        assert torch.equal(res["off"][0][k], res["on"][0][k]), k
    for a, b in zip(res["off"][1], res["on"][1]):
        assert torch.isfinite(b).all()
        assert float((a - b).abs().max()) <= 1e-5 * max(float(a.abs().max()), 1e-30)
    assert float((res["off"][2] - res["on"][2]).abs().max()) <= 1e-5 * float(res["off"][2].abs().max())
    # and straight through the C ABI with debug=1 on a raw call (forward + backward entry points)
    inp = util.scene_inputs(3000, 200, 150, scene_seed=4)
    from diff_gaussian_rasterization import _C
    d = lambda t: t.cuda().contiguous()
    e = torch.Tensor([])
    out = _C.rasterize_gaussians(d(inp["bg"]), d(inp["means3D"]), e, d(inp["opacities"]), d(inp["tongue_class"]),
                                 d(inp["scales"]), d(inp["rotations"]), 1.0, e, d(inp["viewmatrix"]), d(inp["projmatrix"]),
                                 inp["tanfovx"], inp["tanfovy"], 0.0, 150, 200, d(inp["shs"]), 3, d(inp["campos"]), False,
                                 True, True, True)
    out0, _ = util.hip_forward_raw(inp, "TTT")
    assert out[0] == out0[0] and torch.equal(out[1], out0[1]) and torch.equal(out[9], out0[9])
    grads = S.make_upstream_grads(150, 200)
    g1 = util.hip_backward_raw(inp, out, grads, "TTT", debug=True)
    g0 = util.hip_backward_raw(inp, out0, grads, "TTT", debug=False)
    for n in util.GRAD_NAMES:
        assert util.grad_err(g1[n], g0[n]) <= 1e-5, n


@pytest.mark.parametrize("P,variant", [(4001, "TTT"), (4003, "FTT")])
def test_backward_parity_odd_P_poisoned_allocator(P, variant):
    """Odd P: the two gradient-record arrays of the backward workspace are not adjacent (64-byte records, 128-byte
    alignment), and the workspace comes from torch.empty.  The allocator is poisoned first (a block of the same size
    filled with 0xFF = NaN pattern and freed), so any record the backward forgets to zero shows up in Gaussian P-1."""
    _need_gpu()
    from ed3dgs_amd import _lib
    from ed3dgs_amd import synthetic as S
    import ctypes as C
    inp = util.scene_inputs(P, 256, 192, scene_seed=31, kernel_size=0.3)
    grads = util.zero_unused_grads(S.make_upstream_grads(192, 256), variant)
    fw = util.oracle_forward(inp, variant)
    grads, frac = util.mask_marginal(grads, fw, MARGIN)
    out, sv = util.hip_forward_raw(inp, variant)
    bw = util.oracle_backward(inp, util.oracle_state_from_hip(fw, out, sv), grads, variant)
    rc = util.VARIANTS[variant][0]
    nbytes = _lib.lib().ed3dgs_backward_workspace_bytes(C.c_int(P), C.c_int(rc))
    for _ in range(3):
        poison = torch.full((int(nbytes),), 0xFF, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        del poison
        got = util.hip_backward_raw(inp, out, grads, variant)
        for n in util.GRAD_NAMES:
            g = got[n].reshape(bw[n].shape)
            assert np.isfinite(g).all(), n
            assert util.grad_err(g, bw[n]) <= TOL_GRAD, (n, util.grad_err(g, bw[n]))
        # Gaussian P-1 (and P-2) specifically, relative to the whole array's scale
        for n in ("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dmeans2D"):
            g = got[n].reshape(bw[n].shape)
            scale = max(np.abs(bw[n]).max(), 1e-30)
            assert np.abs(g[-2:] - bw[n][-2:]).max() <= TOL_GRAD * scale, n


def test_c4_item_200k_nersemble_shape_deform_on():
    """BASELINE.json configs[3], one item: 200k Gaussians, deformation on (300 timesteps), 1100x1604 portrait
    (NeRSemble 2200x3208 at -r 2), FTT.  (1) deformation vs the numpy restatement pinned by the reference's goldens;
    (2) the rasterizer on the HIP deformation's own outputs vs the (OpenMP) C oracle on those same numbers: tile lists
    bit-exact, images 1e-4; (3) render() itself returns those images bit for bit; (4) backward kernel-level parity and
    the size-independent properties (linearity, zero in -> zero out)."""
    _need_gpu()
    from ed3dgs_amd import synthetic as S
    from ed3dgs_amd.activations import fused_activations
    from ed3dgs_amd.model import PIPE, SynthGaussianModel, default_hyper
    from gaussian_renderer import render
    from oracle import deformation_ref as R
    dev = "cuda"
    P, W, H = 200_000, 1100, 1604
    frames = 300
    t = 137 / frames
    scene = S.make_scene(P, seed=0)
    hy = default_hyper(total_num_frames=frames)
    model = SynthGaussianModel(scene, args=hy, device=dev)
    cam = S.make_cameras(16, W, H, seed=1, device=dev)[5].with_time(t)
    kw = dict(kernel_size=0.0, require_coord=False, require_depth=True, iter=20000, num_down_emb_c=30, num_down_emb_f=30)
    with torch.no_grad():
        m3, sc_f, ro_f, op_f, sh_f, _ = model._deformation(model.get_xyz, model._scaling, model._rotation, model._opacity, t,
                                                          None, model, None, model.get_features, iter=20000,
                                                          num_down_emb_c=30, num_down_emb_f=30)
        sc_a, ro_a, op_a = fused_activations(sc_f, ro_f, op_f, None)
    # (1) deformation
    sd = {k: v.detach().cpu().numpy() for k, v in model._deformation.state_dict().items()}
    fin, _, _ = R.forward(sd, hy, hy.defor_depth, hy.max_embeddings, scene.xyz.numpy(), scene.log_scale.numpy(),
                          scene.rot.numpy(), scene.opacity.numpy(), torch.cat((scene.f_dc, scene.f_rest), 1).numpy(),
                          scene.embedding.numpy(), t, None, 20000, 30, 30)
    for got, ref, name in ((m3, fin[0], "xyz"), (sc_f, fin[1], "scale"), (ro_f, fin[2], "rot"), (op_f, fin[3], "opacity"), (sh_f, fin[4], "sh")):
        err = np.abs(got.cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())
        assert err <= 1e-4, (name, err)
    # (2) rasterizer on identical inputs
    inp = dict(P=P, W=W, H=H, bg=torch.ones(3), means3D=m3.cpu(), opacities=op_a.cpu(), tongue_class=scene.tongue_class,
               scales=sc_a.cpu(), rotations=ro_a.cpu(), shs=sh_f.cpu(), viewmatrix=cam.world_view_transform.cpu(),
               projmatrix=cam.full_proj_transform.cpu(), campos=cam.camera_center.cpu(), tanfovx=math.tan(cam.FoVx * 0.5),
               tanfovy=math.tan(cam.FoVy * 0.5), kernel_size=0.0, scale_modifier=1.0, sh_degree=3)
    fw = util.oracle_forward(inp, "FTT")
    out, sv = util.hip_forward_raw(inp, "FTT")
    _check_state(fw, out, sv)
    errs, frac = _check_images(fw, out, "FTT")
    good = fw["margin"] >= MARGIN
    np.testing.assert_array_equal(sv["n_contrib"][0][good], fw["n_contrib"][0][good])
    print("C4 item: R", fw["num_rendered"], "fwd rel-Linf", errs, "masked px frac %.2e" % frac)
    assert frac <= MAX_MASKED_FRAC
    # (3) render() == the raw call
    pkg = render(cam, model, PIPE, torch.ones(3, device=dev), **kw)
    for k, i in (("render", 1), ("mask", 4), ("normal", 6), ("expected_depth", 7), ("median_depth", 8), ("radii", 9)):
        assert torch.equal(pkg[k].detach(), out[i]), k
    # (4) backward
    grads = util.zero_unused_grads(S.make_upstream_grads(H, W), "FTT")
    grads, _ = util.mask_marginal(grads, fw, MARGIN)
    bw = util.oracle_backward(inp, util.oracle_state_from_hip(fw, out, sv), grads, "FTT")
    got = util.hip_backward_raw(inp, out, grads, "FTT")
    gerr = {n: util.grad_err(got[n].reshape(bw[n].shape), bw[n]) for n in util.GRAD_NAMES}
    print("C4 item bwd (kernel level)", gerr)
    for n, v in gerr.items():
        assert v <= TOL_GRAD, (n, v)
    g1 = {k: v.to(dev) for k, v in S.make_upstream_grads(H, W, seed=3).items()}
    g2 = {k: v.to(dev) for k, v in S.make_upstream_grads(H, W, seed=4).items()}
    params = model.parameters()

    def grads_for(gc, gd):
        for p in params:
            p.grad = None
        pk = render(cam, model, PIPE, torch.ones(3, device=dev), **kw)
        ((pk["render"] * gc).sum() + (pk["expected_depth"] * gd).sum()).backward()
        return [p.grad.detach().clone() for p in params]

    ga, gb = grads_for(g1["color"], g1["depth"]), grads_for(g2["color"], g2["depth"])
    gab = grads_for(g1["color"] + g2["color"], g1["depth"] + g2["depth"])
    gz = grads_for(torch.zeros_like(g1["color"]), torch.zeros_like(g1["depth"]))
    for x, y, z, zero in zip(ga, gb, gab, gz):
        assert float((x + y - z).abs().max()) <= 2e-4 * max(float(z.abs().max()), 1e-30)
        assert float(zero.abs().max()) == 0.0 and torch.isfinite(z).all()
