"""The CPU restatement of the point-integration path (oracle/raster_ref.c: K11-K14, oracle/raster_oracle.integrate)
against answers known in closed form.  The reference ships no fixture for `integrate` and its CUDA extension cannot be
built here: parity unpinned; these cases pin the restatement itself."""
import math

import numpy as np
import torch

import util
from oracle import raster_oracle as O


def _one_gaussian(points, s=0.1, opacity=0.9, W=64, H=48, depth=4.0):
    """One isotropic Gaussian of standard deviation s on the optical axis at `depth`, identity view."""
    fov = 0.6
    tanx = math.tan(fov / 2); tany = tanx * H / W
    view = np.eye(4, dtype=np.float32)                       # row-major transposed W2C = identity
    zn, zf = 0.01, 100.0
    proj = np.zeros((4, 4), np.float32)                      # getProjectionMatrix (utils/graphics_utils.py), transposed
    proj[0, 0] = 1 / tanx; proj[1, 1] = 1 / tany; proj[2, 2] = zf / (zf - zn); proj[2, 3] = 1.0; proj[3, 2] = -(zf * zn) / (zf - zn)
    means = np.array([[0, 0, depth]], np.float32)
    return O.integrate(np.ones(3, np.float32), points, means, np.array([[0.2, 0.5, 0.8]], np.float32),
                       np.array([[opacity]], np.float32), np.full((1, 3), s, np.float32),
                       np.array([[1, 0, 0, 0]], np.float32), 1.0, None, view, proj, tanx, tany, 0.0, H, W, None, 0,
                       np.zeros(3, np.float32))


def test_points_on_the_axis_known_answers():
    s, op, depth = 0.1, 0.9, 4.0
    # the axis passes between pixels; probe the centre of the pixel right of it instead: ray through (px + 0.5, py + 0.5)
    W, H = 64, 48
    fx = W / (2 * math.tan(0.3))
    ray = np.array([0.5 / fx, 0.5 / fx, 1.0])
    ray /= np.linalg.norm(ray)
    t0 = depth / ray[2]                                      # ray distance at which the ray is closest to the centre (approx.)
    dists = np.array([t0 - 1.0, t0 - s, t0, t0 + 1.0])
    pts = (ray[None, :] * dists[:, None]).astype(np.float32)
    r = _one_gaussian(pts, s=s, opacity=op, W=W, H=H, depth=depth)
    assert r["point_valid"].all() and r["condition"][0] == 1
    a = r["alpha_integrated"]
    # Closed form.  The Gaussian's centre projects to ndc2Pix(0) = W/2 - 0.5 (CR/auxiliary.h:40-43) while a query point
    # projects to focal * x / z + W/2 (CR/forward.cu:1061): the probe ray through pixel centre (W/2 + 0.5) is therefore
    # 1 px from the centre on each axis in the kernel's coordinates.  On the axis the inverse ray-space covariance is
    # diag((z / (s f))^2, (z / (s f))^2, 1 / s^2) in (px, px, ray distance).
    assert np.allclose(r["means2D"][0], [W / 2 - 0.5, H / 2 - 0.5]) and np.allclose(r["points2D"][0], [W / 2 + 0.5, H / 2 + 0.5])
    c = (depth / (s * fx)) ** 2
    assert np.allclose(r["invraycov"][0], [c, 0, 0, c, 0, 1 / s ** 2], rtol=1e-4, atol=1e-5)
    peak = op * math.exp(-0.5 * 2 * c)
    dz = 4.0 - (t0 - s)
    assert a[0] == 0.0                                                # 10 sigma in front: alpha < 1/255, skipped
    assert abs(a[1] - op * math.exp(-0.5 * (2 * c + (dz / s) ** 2))) < 1e-4   # one sigma in front
    assert abs(a[2] - peak) < 1e-4 and abs(a[3] - peak) < 1e-4        # at / behind the surface: the full alpha
    # signed distance along the ray: positive in front, negative behind, ~0 at the surface
    sdf = r["sdf"]
    assert abs(sdf[0] - 1.0) < 2e-2 and abs(sdf[1] - s) < 2e-2 and abs(sdf[2]) < 2e-2 and abs(sdf[3] + 1.0) < 2e-2
    # the point's colour is its pixel's rendered colour; the coordinate is its projection
    px, py = int(r["points2D"][0, 0]), int(r["points2D"][0, 1])
    assert (px, py) == (W // 2, H // 2)
    assert np.allclose(r["color_integrated"][0], r["out_color"][:3, py, px])
    assert r["out_color"][8, py, px] == 4.0 and r["out_color"][8].sum() == 4.0   # number of points per pixel


def test_culled_points_keep_their_fill_values():
    pts = np.array([[0, 0, -1.0], [100.0, 0, 4.0], [0, 0, 0.1]], np.float32)    # behind, outside the image, nearer than 0.2
    r = _one_gaussian(pts)
    assert not r["point_valid"].any()
    assert np.array_equal(r["alpha_integrated"], np.ones(3, np.float32))        # DGR/rasterize_points.cu:316
    assert np.array_equal(r["sdf"], np.full(3, -1000.0, np.float32))            # :319
    assert not r["color_integrated"].any() and not r["coordinate2d"].any()


def test_gaussian_state_is_the_rasterizers_and_the_blend_telescopes():
    """K11 is K1 plus the inverse ray-space covariance: everything else it writes, and the tile lists, equal the
    rasterizer's.  The centre-sample blend telescopes: sum_i alpha_i T_i = 1 - T_final."""
    inp = util.scene_inputs(300, 96, 64)
    n = lambda t: t.detach().cpu().numpy()
    fw = util.oracle_forward(inp, "FTT", with_margin=False)
    pts = np.random.default_rng(0).uniform(-1, 1, size=(200, 3)).astype(np.float32) * 2
    r = O.integrate(n(inp["bg"]), pts, n(inp["means3D"]), None, n(inp["opacities"]), n(inp["scales"]), n(inp["rotations"]),
                    1.0, None, n(inp["viewmatrix"]), n(inp["projmatrix"]), inp["tanfovx"], inp["tanfovy"], 0.0, 64, 96,
                    n(inp["shs"]), 3, n(inp["campos"]))
    assert r["num_rendered"] == fw["num_rendered"] and np.array_equal(r["point_list"], fw["point_list"])
    for k in ("means2D", "conic_opacity", "ts", "ray_planes", "rgb", "radii"):
        assert np.array_equal(r[k], fw[k]), k
    assert np.abs(r["out_color"][7] + r["accum_alpha"][0] - 1.0).max() < 1e-5
    v = r["point_valid"]
    assert v.any() and (r["alpha_integrated"][v] >= 0).all() and (r["alpha_integrated"][v] <= 1 + 1e-6).all()
    assert r["out_color"][8].sum() == v.sum()
    # symmetric inverse covariance of a well-conditioned Gaussian is positive semi-definite along the ray
    wc = r["condition"].astype(bool) & (r["radii"] > 0)
    assert wc.any() and (r["invraycov"][wc][:, 5] > 0).all()
