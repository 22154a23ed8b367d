"""The point-integration path on the GPU (csrc/integrate.hip + the INTE preprocess, through _C.integrate_gaussians_to_points
and GaussianRasterizer.integrate) against the CPU restatement (oracle/raster_oracle.integrate).  Tolerance 1e-4 relative
(L-inf, per output); pixels / points whose threshold decisions (alpha >= 1/255, T >= 1e-4, T > 0.5, point depth vs plane
depth) lie within 1e-3 of the threshold are left out -- a last-ulp difference of exp() flips them -- and must be few."""
import numpy as np
import pytest
import torch

import util
from oracle import raster_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU test collected without a GPU")


def _points(inp, n, seed):
    """Query points spread through the scene's depth range along random pixel rays + some culled ones."""
    g = np.random.default_rng(seed)
    view = inp["viewmatrix"].numpy().T          # W2C
    c2w = np.linalg.inv(view)
    W, H = inp["W"], inp["H"]
    fx, fy = W / (2 * inp["tanfovx"]), H / (2 * inp["tanfovy"])
    u, v = g.uniform(-2, W + 2, n), g.uniform(-2, H + 2, n)
    z = g.uniform(0.1, 12.0, n)
    cam = np.stack([(u - W / 2) / fx * z, (v - H / 2) / fy * z, z, np.ones(n)], 1)
    cam[: n // 20, 2] *= -1                     # some behind the camera
    return (cam @ c2w.T)[:, :3].astype(np.float32)


def _compare(ref, got, what, mask=None):
    a, b = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    if mask is not None:
        a, b = a[mask], b[mask]
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30) if a.size else 0.0
    assert err <= TOL, (what, err)
    return err


@pytest.mark.parametrize("P,W,H,PN,colors", [(400, 96, 64, 3000, False), (2000, 200, 120, 20000, False), (900, 131, 77, 5000, True)],
                         ids=["small", "medium", "ragged-precomp-colors"])
def test_integrate_matches_restatement(P, W, H, PN, colors):
    _need_gpu()
    from diff_gaussian_rasterization import _C
    inp = util.scene_inputs(P, W, H)
    pts = _points(inp, PN, seed=P)
    n = lambda t: t.detach().cpu().numpy()
    cp = torch.rand(P, 3, generator=torch.Generator().manual_seed(3)) if colors else None
    ref = O.integrate(n(inp["bg"]), pts, n(inp["means3D"]), None if cp is None else n(cp), n(inp["opacities"]), n(inp["scales"]),
                      n(inp["rotations"]), 1.0, None, n(inp["viewmatrix"]), n(inp["projmatrix"]), inp["tanfovx"],
                      inp["tanfovy"], 0.0, H, W, None if colors else n(inp["shs"]), 3, n(inp["campos"]))
    d = lambda t: t.cuda().contiguous()
    e = torch.Tensor([])
    _C.KEEP_LAST = True
    try:
        out = _C.integrate_gaussians_to_points(
            d(inp["bg"]), torch.from_numpy(pts).cuda(), d(inp["means3D"]), d(cp) if colors else e, d(inp["opacities"]),
            d(inp["scales"]), d(inp["rotations"]), 1.0, e, e, d(inp["viewmatrix"]), d(inp["projmatrix"]), inp["tanfovx"],
            inp["tanfovy"], 0.0, None, H, W, e if colors else d(inp["shs"]), 3, d(inp["campos"]), False, False)
        L = dict(_C.LAST)
    finally:
        _C.KEEP_LAST = False
        _C.LAST.clear()
    rendered, color, a_int, c_int, coord, sdf, radii = [x.cpu().numpy() if torch.is_tensor(x) else x for x in out[:7]]
    assert rendered == ref["num_rendered"]
    assert np.array_equal(radii, ref["radii"])
    # K11: condition flags and inverse ray-space covariances
    vis = ref["radii"] > 0
    assert np.array_equal(L["condition"].cpu().numpy()[vis], ref["condition"][vis])
    inv = L["invraycov"].cpu().numpy()
    scale = np.abs(ref["invraycov"][vis]).max(1, keepdims=True) + 1e-30
    assert (np.abs(inv[vis] - ref["invraycov"][vis]) / scale).max() < 1e-3     # eigen-decomposition based: ~1e-5 typical
    # image part
    ok_pix = ref["pix_margin"] > 1e-3
    assert ok_pix.mean() > 0.9
    for ch, name in ((0, "r"), (1, "g"), (2, "b"), (3, "expected distance"), (4, "median distance"), (6, "max distance"), (7, "alpha")):
        _compare(ref["out_color"][ch], color[ch], "image " + name, ok_pix)
    assert np.array_equal(color[8], ref["out_color"][8])                        # points per pixel: exact
    assert not color[5].any()
    _compare(ref["accum_alpha"][0], L["accum_alpha"].cpu().numpy()[0], "final T", ok_pix)
    # point part
    valid = ref["point_valid"]
    assert 0.5 < valid.mean() < 1.0
    assert np.array_equal(a_int[~valid], np.ones((~valid).sum(), np.float32)) and np.array_equal(sdf[~valid], np.full((~valid).sum(), -1000.0, np.float32))
    assert not c_int[~valid].any() and not coord[~valid].any()
    assert np.array_equal(coord[valid], ref["coordinate2d"][valid])             # projections: bit-exact
    ok_pt = valid & (ref["pt_margin"] > 1e-3)
    assert ok_pt.sum() > 0.85 * valid.sum()
    print("alpha_integrated", _compare(ref["alpha_integrated"], a_int, "alpha_integrated", ok_pt),
          "sdf", _compare(ref["sdf"], sdf, "sdf", ok_pt), "color", _compare(ref["color_integrated"], c_int, "color_integrated", ok_pt),
          "compared points", int(ok_pt.sum()), "of", int(valid.sum()))


def test_more_than_256_points_in_one_pixel_and_empty_inputs():
    """The reference batches a pixel's points by MAX_NUM_PROJECTED = 256; the result per point must not depend on it."""
    _need_gpu()
    from diff_gaussian_rasterization import _C
    inp = util.scene_inputs(300, 64, 48)
    g = np.random.default_rng(5)
    view = inp["viewmatrix"].numpy().T
    c2w = np.linalg.inv(view)
    fx, fy = 64 / (2 * inp["tanfovx"]), 48 / (2 * inp["tanfovy"])
    z = g.uniform(0.5, 10.0, 700)
    u, v = 30.25 + g.uniform(0, 0.5, 700), 20.25 + g.uniform(0, 0.5, 700)       # all inside pixel (30, 20)
    cam = np.stack([(u - 32) / fx * z, (v - 24) / fy * z, z, np.ones(700)], 1)
    pts = (cam @ c2w.T)[:, :3].astype(np.float32)
    n = lambda t: t.detach().cpu().numpy()
    ref = O.integrate(n(inp["bg"]), pts, n(inp["means3D"]), None, n(inp["opacities"]), n(inp["scales"]), n(inp["rotations"]),
                      1.0, None, n(inp["viewmatrix"]), n(inp["projmatrix"]), inp["tanfovx"], inp["tanfovy"], 0.0, 48, 64,
                      n(inp["shs"]), 3, n(inp["campos"]))
    d = lambda t: t.cuda().contiguous()
    e = torch.Tensor([])
    args = lambda p: (d(inp["bg"]), p, d(inp["means3D"]), e, d(inp["opacities"]), d(inp["scales"]), d(inp["rotations"]), 1.0, e, e,
                      d(inp["viewmatrix"]), d(inp["projmatrix"]), inp["tanfovx"], inp["tanfovy"], 0.0, None, 48, 64, d(inp["shs"]), 3,
                      d(inp["campos"]), False, False)
    out = _C.integrate_gaussians_to_points(*args(torch.from_numpy(pts).cuda()))
    assert float(out[1][8, 20, 30]) == 700.0
    ok = ref["point_valid"] & (ref["pt_margin"] > 1e-3)
    assert ok.sum() > 500
    _compare(ref["alpha_integrated"], out[2].cpu().numpy(), "alpha_integrated", ok)
    _compare(ref["sdf"], out[5].cpu().numpy(), "sdf", ok)
    # no points / no Gaussians: fill values, num_rendered 0 (DGR/rasterize_points.cu:345)
    out0 = _C.integrate_gaussians_to_points(*args(torch.zeros(0, 3).cuda()))
    assert out0[0] == 0 and out0[2].shape == (0,) and not out0[1].any()


def test_renderer_level_integrate_runs_on_a_model():
    _need_gpu()
    from ed3dgs_amd import synthetic as S
    from ed3dgs_amd.model import PIPE, SynthGaussianModel, default_hyper
    from gaussian_renderer import integrate
    model = SynthGaussianModel(S.make_scene(3000, seed=0), args=default_hyper(), device="cuda")
    cam = S.make_cameras(2, 160, 120, seed=1, device="cuda")[1].with_time(0.3)
    pts = model.get_xyz.detach()[:1500] + 0.05 * torch.randn(1500, 3, device="cuda")
    r = integrate(pts, cam, model, PIPE, torch.ones(3, device="cuda"), 0.0, 20000, num_down_emb_c=30, num_down_emb_f=30)
    assert set(r) == {"render", "alpha_integrated", "color_integrated", "point_coordinate", "point_sdf", "visibility_filter", "radii"}
    assert r["render"].shape == (9, 120, 160) and r["alpha_integrated"].shape == (1500,)
    a = r["alpha_integrated"]
    assert torch.isfinite(a).all() and float(a.min()) >= 0 and float(a.max()) <= 1 + 1e-5
    inside = r["point_coordinate"].abs().sum(1) > 0
    assert inside.float().mean() > 0.3 and float(r["render"][8].sum()) == float(inside.sum())
    with pytest.raises(Exception):
        from diff_gaussian_rasterization import GaussianRasterizer
        GaussianRasterizer(None).integrate(pts, pts, pts, pts)      # neither SHs nor colours


def test_no_visible_gaussians():
    """Every Gaussian behind the camera: num_rendered 0, the image is the background, valid points integrate nothing."""
    _need_gpu()
    from diff_gaussian_rasterization import _C
    inp = util.scene_inputs(200, 64, 48)
    view = inp["viewmatrix"].numpy().T
    c2w = np.linalg.inv(view)
    behind = (np.concatenate([np.random.default_rng(0).uniform(-1, 1, (200, 2)), -np.full((200, 1), 5.0), np.ones((200, 1))], 1) @ c2w.T)[:, :3]
    pts = (np.array([[0.0, 0.0, 3.0, 1.0], [0.2, -0.1, 6.0, 1.0]]) @ c2w.T)[:, :3].astype(np.float32)
    d = lambda t: t.cuda().contiguous()
    e = torch.Tensor([])
    out = _C.integrate_gaussians_to_points(
        d(inp["bg"]), torch.from_numpy(pts).cuda(), torch.from_numpy(behind.astype(np.float32)).cuda(), e, d(inp["opacities"]),
        d(inp["scales"]), d(inp["rotations"]), 1.0, e, e, d(inp["viewmatrix"]), d(inp["projmatrix"]), inp["tanfovx"],
        inp["tanfovy"], 0.0, None, 48, 64, d(inp["shs"]), 3, d(inp["campos"]), False, False)
    assert out[0] == 0 and not out[6].any()
    assert torch.equal(out[1][:3], torch.ones(3, 48, 64, device="cuda")) and not out[1][3:8].any()
    assert float(out[1][8].sum()) == 2.0
    assert torch.equal(out[2], torch.zeros(2, device="cuda")) and torch.equal(out[3], torch.ones(2, 3, device="cuda"))
    assert (out[5] < 0).all()          # sdf = 0 - depth: no surface in front of the point
