"""GPU parity for the reference paths no earlier test executed (VERDICT r3 "missing" 2-4):

  (a) active_sh_degree 0, 1, 2 with the M = 16 coefficient rows the reference keeps while the degree grows
      (scene/gaussian_model.py:49,146-148, train.py:129-130; CR/forward.cu:23-74, CR/backward.cu:21-140): forward (rgb, clamp
      mask, images, lists) and backward (kernel level; rows above (deg+1)^2 of dL_dsh EXACTLY zero);
  (b) the backward with colors_precomp + cov3D_precomp (dL_dcolors, dL_dcov3D as leaf outputs), and once through
      render(override_color=...) and pipe.compute_cov3D_python / pipe.convert_SHs_python
      (gaussian_renderer/__init__.py:68-72,85-95);
  (c) scale_modifier = 0.7 forward + backward (CR/forward.cu:270-304, CR/backward.cu:492-555, quirk Q14);
  (d) the reference caller's argument shapes: six 0-d GPU tensors in the settings and a (P,1) GPU time tensor
      (gaussian_renderer/__init__.py:29-37,45) -> bit-identical to the plain-number call;
  (e) the backward with a background that is not white (it enters K7's suffix sum);
  (f) the HIP state's rgb / clamp mask / cov3D, the torch eval_sh of the convert_SHs_python branch, the 3D filter and the fused
      3D-filter activations against the fixtures generated from the REFERENCE's own Python (tools/gen_raster_golden.py) --
      directly, not through the oracle.
Tolerances are the existing ones (test_raster_parity_gpu.py): lists bit-exact, images 1e-4, kernel-level gradients 5e-5.
"""
import math
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import util
from test_raster_parity_gpu import MARGIN, MAX_MASKED_FRAC, TOL_GRAD, TOL_GRAD_E2E, TOL_IMG, _check_images, _check_state

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _fwd_bwd(inp, variant, colors=None, cov=None, e2e=True, grad_seed=3):
    """Forward state + images vs the oracle, then the backward at kernel level (oracle backward fed the HIP forward's saved
    state) and end to end.  Returns (fw, out, sv, hip gradients, oracle kernel-level gradients)."""
    from ed3dgs_amd import synthetic as S
    fw = util.oracle_forward(inp, variant, colors_precomp=colors, cov3D_precomp=cov)
    out, sv = util.hip_forward_raw(inp, variant, colors_precomp=colors, cov3D_precomp=cov)
    _check_state(fw, out, sv, colors, cov)
    _check_images(fw, out, variant)
    grads = util.zero_unused_grads(S.make_upstream_grads(inp["H"], inp["W"], seed=grad_seed), variant)
    grads, frac = util.mask_marginal(grads, fw, MARGIN)
    assert frac <= MAX_MASKED_FRAC
    bw = util.oracle_backward(inp, util.oracle_state_from_hip(fw, out, sv), grads, variant, colors_precomp=colors, cov3D_precomp=cov)
    got = util.hip_backward_raw(inp, out, grads, variant, colors_precomp=colors, cov3D_precomp=cov)
    errs = {}
    for n in util.GRAD_NAMES:
        if (n == "dL_dsh" and colors is not None) or (n in ("dL_dscales", "dL_drotations") and cov is not None):
            assert not np.asarray(got[n]).any(), n                   # no such input: the slot stays zero
            continue
        a = got[n].reshape(np.asarray(bw[n]).shape)
        assert np.isfinite(a).all(), n
        errs[n] = util.grad_err(a, bw[n])
    print(variant, "bwd kernel-level rel-Linf", {k: "%.1e" % v for k, v in errs.items()})
    for n, v in errs.items():
        assert v <= TOL_GRAD, (n, v, errs)
    if e2e:
        bw2 = util.oracle_backward(inp, fw, grads, variant, colors_precomp=colors, cov3D_precomp=cov)
        for n in errs:
            v = util.grad_err(got[n].reshape(np.asarray(bw2[n]).shape), bw2[n])
            assert v <= TOL_GRAD_E2E, ("end to end", n, v)
    return fw, out, sv, got, bw


# ------------------------------------------------------------------ (a) lower SH degrees
@pytest.mark.parametrize("deg", [0, 1, 2])
@pytest.mark.parametrize("variant", ["FFF", "FTT"])
def test_lower_sh_degrees_c1(deg, variant):
    _need_gpu()
    inp = util.scene_inputs(10000, 400, 400, tongue=True, sh_degree=deg)
    assert inp["shs"].shape[1] == 16
    fw, out, sv, got, bw = _fwd_bwd(inp, variant)
    vis = fw["radii"] > 0
    np.testing.assert_array_equal(sv["clamped"][vis], _clamp_byte(fw["clamped"])[vis])
    n = (deg + 1) ** 2
    dsh = got["dL_dsh"].reshape(inp["P"], 16, 3)
    assert np.abs(dsh[:, :n]).max() > 0
    assert not dsh[:, n:].any(), "coefficient rows above the active degree must receive an exact zero"
    # the view direction's share of dL_dmeans3D exists from degree 1 on (CR/backward.cu:44-140): included in the comparison
    # above; here only that the degrees differ where they should
    if deg == 0:
        inp3 = util.scene_inputs(10000, 400, 400, tongue=True, sh_degree=3)
        assert np.abs(util.oracle_forward(inp3, variant, with_margin=False)["rgb"] - fw["rgb"]).max() > 1e-3


def _clamp_byte(clamped):
    c = np.asarray(clamped).reshape(-1, 3).astype(np.uint8)
    return c[:, 0] | (c[:, 1] << 1) | (c[:, 2] << 2)


# ------------------------------------------------------------------ (b) precomputed colour / covariance backward
@pytest.mark.parametrize("variant,ks", [("FTT", 0.0), ("TTT", 0.3)])
def test_backward_with_precomputed_colors_and_cov3d(variant, ks):
    _need_gpu()
    inp = util.scene_inputs(10000, 400, 400, scene_seed=5, kernel_size=ks)
    g = torch.Generator().manual_seed(11)
    colors = torch.rand(inp["P"], 3, generator=g)
    cov = torch.from_numpy(util.oracle_forward(inp, "FFF", with_margin=False)["cov3D"].copy())
    fw, out, sv, got, bw = _fwd_bwd(inp, variant, colors=colors, cov=cov)
    assert np.abs(got["dL_dcolors"]).max() > 0 and np.abs(got["dL_dcov3D"]).max() > 0
    # colours only / covariance only: the two switches are independent (CR/forward.cu:484,528)
    _fwd_bwd(inp, variant, colors=colors, e2e=False)
    _fwd_bwd(inp, variant, cov=cov, e2e=False)


def _static_model(P, seed, sh_active=3):
    from ed3dgs_amd import synthetic as S
    from ed3dgs_amd.model import SynthGaussianModel, default_hyper
    sc = S.make_scene(P, seed=seed)
    m = SynthGaussianModel(sc, args=default_hyper(no_coarse_deform=True, no_fine_deform=True), device="cuda")
    m.active_sh_degree = sh_active
    return sc, m


def _render_inputs(sc, cam, W, H, ks, mod=1.0, bg=(1.0, 1.0, 1.0), deg=3):
    from ed3dgs_amd import synthetic as S
    a = S.activated(sc)
    return dict(P=sc.xyz.shape[0], W=W, H=H, bg=torch.tensor(bg), means3D=sc.xyz, opacities=a["opacities"],
                tongue_class=sc.tongue_class, scales=a["scales"], rotations=a["rotations"], shs=a["shs"],
                viewmatrix=cam.world_view_transform.cpu(), projmatrix=cam.full_proj_transform.cpu(),
                campos=cam.camera_center.cpu(), tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
                kernel_size=ks, scale_modifier=mod, sh_degree=deg)


def _render_and_state(fn):
    from diff_gaussian_rasterization import _C
    _C.KEEP_LAST = True
    try:
        pkg = fn()
        L = dict(_C.LAST)
        sv = _C.state_view(L["P"], L["H"], L["W"], L["R"], L["geom"], L["binning"], L["img"])
    finally:
        _C.KEEP_LAST = False
        _C.LAST.clear()
    return pkg, L, sv


def _pkg_out(pkg, L):
    out = [L["R"], pkg["render"], pkg["expected_coord"], pkg["median_coord"], pkg["mask"], pkg["tongue_mask"], pkg["normal"],
           pkg["expected_depth"], pkg["median_depth"]]
    return [o.detach() if torch.is_tensor(o) else o for o in out]


def _backward_through(pkg, grads):
    outs = [pkg["render"], pkg["expected_coord"], pkg["median_coord"], pkg["expected_depth"], pkg["median_depth"], pkg["mask"],
            pkg["normal"]]
    ups = [grads[k].cuda() for k in ("color", "coord", "mcoord", "depth", "mdepth", "alpha", "normal")]
    torch.autograd.backward(outs, ups)


def test_render_override_color_forward_and_leaf_gradient():
    """render(override_color=c): the rasterizer takes c as colors_precomp (no SH), images equal the oracle's, and c -- a leaf --
    receives dL_dcolors.  (The reference passes BOTH shs and colors_precomp on this branch, :96-104, and its rasterizer then
    raises "Please provide excatly one of either SHs or precomputed colors!"; here the branch does what it is written for.)"""
    _need_gpu()
    from ed3dgs_amd import synthetic as S
    from ed3dgs_amd.model import PIPE
    import gaussian_renderer as GR
    P, W, H = 6000, 320, 240
    sc, model = _static_model(P, seed=23)
    cam = S.make_cameras(3, W, H, seed=5, device="cuda")[2].with_time(0.5)
    g = torch.Generator().manual_seed(12)
    colors = torch.rand(P, 3, generator=g)
    c_leaf = colors.cuda().requires_grad_(True)
    bg = torch.tensor([0.1, 0.2, 0.3])
    pkg, L, sv = _render_and_state(lambda: GR.render(cam, model, PIPE, bg.cuda(), kernel_size=0.0, require_coord=False,
                                                     require_depth=True, override_color=c_leaf, iter=20000))
    inp = _render_inputs(sc, cam, W, H, 0.0, bg=(0.1, 0.2, 0.3))
    fw = util.oracle_forward(inp, "FTT", colors_precomp=colors)
    np.testing.assert_array_equal(sv["point_list"], fw["point_list"])
    _check_images(fw, _pkg_out(pkg, L), "FTT")
    grads = util.zero_unused_grads(S.make_upstream_grads(H, W, seed=8), "FTT")
    grads, _ = util.mask_marginal(grads, fw, MARGIN)
    _backward_through(pkg, grads)
    fake = [None] * 13
    fake[4], fake[6] = pkg["mask"].detach(), pkg["normal"].detach()
    bw = util.oracle_backward(inp, util.oracle_state_from_hip(fw, fake, sv), grads, "FTT", colors_precomp=colors)
    assert util.grad_err(c_leaf.grad.cpu().numpy(), bw["dL_dcolors"]) <= TOL_GRAD
    assert model._features_dc.grad is None or not model._features_dc.grad.any()      # the SH path was not taken
    assert util.grad_err(model._xyz.grad.cpu().numpy(), bw["dL_dmeans3D"]) <= TOL_GRAD


def test_render_compute_cov3d_python_and_convert_shs_python():
    """pipe.compute_cov3D_python: cov3D_precomp = pc.get_covariance(scaling_modifier) of the model's parameters
    (scene/gaussian_model.py:31-35,143-144) -> rasterizer; its leaf-output gradient dL_dcov3D flows back into _scaling and
    _rotation through torch.  pipe.convert_SHs_python: colours from eval_sh at the active degree.  Both against the oracle fed the
    same precomputed arrays, and the torch chain from dL_dcov3D against autograd of the same formula in float64."""
    _need_gpu()
    from ed3dgs_amd import synthetic as S
    import gaussian_renderer as GR
    P, W, H = 6000, 320, 240
    sc, model = _static_model(P, seed=29, sh_active=2)
    cam = S.make_cameras(3, W, H, seed=6, device="cuda")[1].with_time(0.1)
    pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=True, debug=False)
    mod = 0.7
    pkg, L, sv = _render_and_state(lambda: GR.render(cam, model, pipe, torch.ones(3).cuda(), kernel_size=0.0, scaling_modifier=mod,
                                                     require_coord=False, require_depth=True, iter=20000))
    inp = _render_inputs(sc, cam, W, H, 0.0, mod=mod, deg=2)
    fw0 = util.oracle_forward(inp, "FTT", with_margin=False)       # the oracle's own cov3D and SH colours at modifier 0.7, degree 2
    cov = model.get_covariance(mod).detach().cpu()
    assert (np.abs(cov.numpy() - fw0["cov3D"]).max(1) <= 2e-6 * np.abs(fw0["cov3D"]).max(1)).all()
    colors = torch.from_numpy(sv["rec"][:, 6:9].copy())               # what render() handed over as colors_precomp ...
    vis = fw0["radii"] > 0
    assert np.abs(colors.numpy()[vis] - fw0["rgb"][vis]).max() <= 5e-7  # ... equals the rasterizer's SH path at degree 2
    fw = util.oracle_forward(inp, "FTT", colors_precomp=colors, cov3D_precomp=cov)
    np.testing.assert_array_equal(pkg["radii"].cpu().numpy(), fw["radii"])
    np.testing.assert_array_equal(sv["point_list"], fw["point_list"])
    _check_images(fw, _pkg_out(pkg, L), "FTT")
    grads = util.zero_unused_grads(S.make_upstream_grads(H, W, seed=9), "FTT")
    grads, _ = util.mask_marginal(grads, fw, MARGIN)
    _backward_through(pkg, grads)
    fake = [None] * 13
    fake[4], fake[6] = pkg["mask"].detach(), pkg["normal"].detach()
    bw = util.oracle_backward(inp, util.oracle_state_from_hip(fw, fake, sv), grads, "FTT", colors_precomp=colors, cov3D_precomp=cov)
    # dL_dcov3D -> (_scaling, _rotation) in float64 autograd of the reference's formula
    s = sc.log_scale.double().requires_grad_(True)
    q = sc.rot.double().requires_grad_(True)
    qn = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = qn[:, 0], qn[:, 1], qn[:, 2], qn[:, 3]
    R = torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z),
                     2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)), 1).reshape(-1, 3, 3)
    Lm = R * (mod * torch.exp(s))[:, None, :]
    c = Lm @ Lm.transpose(1, 2)
    c6 = torch.stack((c[:, 0, 0], c[:, 0, 1], c[:, 0, 2], c[:, 1, 1], c[:, 1, 2], c[:, 2, 2]), 1)
    (c6 * torch.from_numpy(np.asarray(bw["dL_dcov3D"], np.float64))).sum().backward()
    assert util.grad_err(model._scaling.grad.cpu().numpy(), s.grad.numpy()) <= 2 * TOL_GRAD
    assert util.grad_err(model._rotation.grad.cpu().numpy(), q.grad.numpy()) <= 2 * TOL_GRAD
    assert util.grad_err(model._xyz.grad.cpu().numpy()[vis], np.asarray(bw["dL_dmeans3D"])[vis]) <= 1e-3   # + the SH direction term via torch


# ------------------------------------------------------------------ (c) scale modifier, (e) background
@pytest.mark.parametrize("variant,ks", [("FTT", 0.0), ("TTT", 0.3)])
def test_scale_modifier_and_coloured_background(variant, ks):
    _need_gpu()
    inp = util.scene_inputs(10000, 400, 400, scene_seed=19, kernel_size=ks, scale_modifier=0.7, bg=(0.1, 0.2, 0.3))
    fw, out, sv, got, bw = _fwd_bwd(inp, variant)
    fw1 = util.oracle_forward({**inp, "scale_modifier": 1.0}, variant, with_margin=False)
    assert fw["num_rendered"] < fw1["num_rendered"]                  # smaller splats: the modifier reached the covariance
    # quirk Q14 (CR/backward.cu:514,540-542): dL_dscales is dL/d(mod * scale) -- compare with the modifier-1 formula on the scaled scale
    assert np.abs(got["dL_dscales"]).max() > 0
    # the background: same scene on white differs in the colour plane and in the gradients K7's suffix sum feeds
    inpw = {**inp, "bg": torch.ones(3)}
    outw, svw = util.hip_forward_raw(inpw, variant)
    assert float((outw[1] - out[1]).abs().max()) > 0.1
    np.testing.assert_array_equal(svw["point_list"], sv["point_list"])


def test_backward_with_coloured_background_c1_all_variants_kernel_level():
    _need_gpu()
    for variant in ("FFF", "TFT"):
        inp = util.scene_inputs(10000, 400, 400, scene_seed=2, bg=(0.9, 0.0, 0.35))
        _fwd_bwd(inp, variant, e2e=False)


# ------------------------------------------------------------------ (d) the reference caller's argument shapes
def test_zero_dim_gpu_tensor_settings_and_p1_time_tensor_are_bit_identical():
    _need_gpu()
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from ed3dgs_amd import synthetic as S
    from ed3dgs_amd.model import SynthGaussianModel, default_hyper
    inp = util.scene_inputs(4000, 320, 240, scene_seed=31, kernel_size=0.1)
    plain = util.hip_settings(inp, "TTT")
    t = lambda v: torch.tensor(v).cuda()
    # gaussian_renderer/__init__.py:28-44: image_height, image_width, tanfovx, tanfovy, scale_modifier, sh_degree as 0-d tensors
    as_tensors = plain._replace(image_height=t(inp["H"]), image_width=t(inp["W"]), tanfovx=t(inp["tanfovx"]), tanfovy=t(inp["tanfovy"]),
                                scale_modifier=t(inp["scale_modifier"]), sh_degree=t(inp["sh_degree"]))
    assert as_tensors.tanfovx.dim() == 0 and as_tensors.tanfovx.is_cuda
    res = []
    for rs in (plain, as_tensors):
        leaf = lambda a: a.cuda().clone().requires_grad_(True)
        m3, op, sc, ro, sh = (leaf(inp[k]) for k in ("means3D", "opacities", "scales", "rotations", "shs"))
        m2 = torch.zeros_like(m3, requires_grad=True)
        outs = GaussianRasterizer(rs)(means3D=m3, means2D=m2, opacities=op, tongue_class=inp["tongue_class"].cuda(), shs=sh,
                                      scales=sc, rotations=ro)
        (outs[0].sum() + outs[4].sum() + outs[8].sum() + outs[2].sum()).backward()
        res.append([o.detach().clone() for o in outs] + [m3.grad, sc.grad, ro.grad, sh.grad, op.grad, m2.grad])
    for a, b in zip(res[0][:9], res[1][:9]):
        assert torch.equal(a, b)                                        # forward: bit-identical
    for a, b in zip(res[0][9:], res[1][9:]):
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max())  # float atomics: summation order only (1.6e-5 seen)
    # the deformation network with the reference's (P,1) GPU time tensor (:45) vs the Python float render() passes
    P = 3000
    scn = S.make_scene(P, seed=4)
    model = SynthGaussianModel(scn, args=default_hyper(), device="cuda")
    feats = model.get_features
    with torch.no_grad():
        a = model._deformation(model._xyz, model._scaling, model._rotation, model._opacity, 0.37, None, model, None, feats,
                               iter=20000, num_down_emb_c=30, num_down_emb_f=30)
        tt = torch.tensor(0.37).cuda().repeat(P, 1)
        b = model._deformation(model._xyz, model._scaling, model._rotation, model._opacity, tt, None, model, None, feats,
                               iter=20000, num_down_emb_c=30, num_down_emb_f=30)
    assert tt.shape == (P, 1)
    for x, y in zip(a[:5], b[:5]):
        assert torch.equal(x, y)


# ------------------------------------------------------------------ (f) HIP vs the reference-generated fixtures, directly
@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_hip_sh_colours_and_clamp_mask_match_reference_fixture(deg):
    _need_gpu()
    import test_raster_golden_cpu as G
    from ed3dgs_amd.sh import eval_sh
    inp, g = G.sh_fixture_inputs(deg)
    out, sv = util.hip_forward_raw(inp, "FFF")
    assert (out[9] > 0).all()
    want = g["rgb_deg%d" % deg]
    err = np.abs(sv["rec"][:, 6:9] - want).max()
    print("degree", deg, "HIP rgb vs the reference's eval_sh: %.2e" % err)
    assert err <= G.TOL
    differ = sv["clamped"] != G.clamp_bits(g["clamped_deg%d" % deg])
    assert differ.mean() < 1e-2 and (np.maximum(sv["rec"][:, 6:9], want)[differ] <= G.TOL).all()
    # the torch polynomials render() falls back to for convert_SHs_python outside the reference's tree
    shs = inp["shs"].cuda()
    d = inp["means3D"].cuda() - inp["campos"].cuda()[None]
    got = torch.clamp_min(eval_sh(deg, shs.transpose(1, 2).reshape(-1, 3, 16), d / d.norm(dim=1, keepdim=True)) + 0.5, 0.0)
    assert np.abs(got.cpu().numpy() - want).max() <= G.TOL


@pytest.mark.parametrize("mod", [1.0, 0.7])
def test_hip_cov3d_matches_reference_fixture(mod):
    _need_gpu()
    import test_raster_golden_cpu as G
    from ed3dgs_amd.model import SynthGaussianModel
    from ed3dgs_amd import synthetic as S
    inp, want = G.cov_fixture_inputs(mod)
    out, sv = util.hip_forward_raw(inp, "FFF")
    err = G.cov_err(sv["cov3D"], want)
    print("modifier", mod, "HIP cov3D vs the reference's build_scaling_rotation: %.2e" % err)
    assert err <= G.TOL
    model = SynthGaussianModel(S.make_scene(want.shape[0], seed=41), device="cuda")
    with torch.no_grad():   # the fixture's activated scales, exactly (exp differs in the last bit between hosts / devices)
        model._scaling.copy_(torch.log(inp["scales"].double()).float().cuda())
        model._rotation.copy_(inp["rotations"].cuda())
    assert G.cov_err(model.get_covariance(mod).detach().cpu().numpy(), want) <= 2 * G.TOL   # the compute_cov3D_python formula (exp(log s) is within an ulp of s)


def test_hip_filter3d_and_filtered_activations_match_reference_fixture():
    _need_gpu()
    import test_raster_golden_cpu as G
    from ed3dgs_amd.activations import fused_activations
    from ed3dgs_amd.filter3d import compute_3D_filter
    g = G._gold("raster_filter3d.npz")
    filt = compute_3D_filter(torch.from_numpy(g["xyz"]).cuda(), G._filter_cams(g))
    assert np.abs(filt.cpu().numpy() - g["filter_3D"]).max() <= 2e-7 * np.abs(g["filter_3D"]).max()
    P = g["xyz"].shape[0]
    rot = torch.nn.functional.normalize(torch.randn(P, 4, generator=torch.Generator().manual_seed(1))).cuda()
    s, r, o = fused_activations(torch.from_numpy(g["log_scale"]).cuda(), rot, torch.from_numpy(g["opacity_logit"]).cuda(),
                                torch.from_numpy(g["filter_3D"]).cuda())
    assert np.abs(s.cpu().numpy() - g["scales_filtered"]).max() <= 2e-6 * np.abs(g["scales_filtered"]).max()
    assert np.abs(o.cpu().numpy() - g["opacity_filtered"]).max() <= 2e-6
