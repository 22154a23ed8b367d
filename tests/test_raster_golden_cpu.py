"""The raster oracle's K1 stages against fixtures generated from the REFERENCE's own Python (tools/gen_raster_golden.py):
eval_sh (utils/sh_utils.py:57), build_scaling_rotation / strip_symmetric (utils/general_utils.py:78-112, composed as
scene/gaussian_model.py:31-35), the camera matrices (utils/graphics_utils.py:106-141, scene/cameras.py:84-92), and
GaussianModel.compute_3D_filter / apply_scaling_n_opacity_with_3D_filter (scene/gaussian_model.py:538-603).

This is the most the reference can pin of the rasterizer here (its CUDA kernels cannot run, it ships no raster fixtures):
SH -> RGB with the clamp mask, cov3D at scale_modifier 1 and 0.7, viewmatrix / projmatrix / campos.  What stays unpinned by
the reference: cov2D / conic / planes / normals / radii (K1's second half), binning, K6, K7, K8+K9."""
import math
import os

import numpy as np
import pytest
import torch

import util
from ed3dgs_amd import synthetic as S

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 5e-7      # VERDICT r3 measured 1.2e-7 (rgb) and 3.8e-7 (cov3D)


def _gold(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def sh_fixture_inputs(deg):
    g = _gold("raster_k1_sh.npz")
    inp = util.scene_inputs(g["means3D"].shape[0], 400, 400, scene_seed=41, sh_degree=deg)
    assert np.array_equal(inp["means3D"].numpy(), g["means3D"])            # scene 41 IS the fixture's scene
    inp["shs"] = torch.from_numpy(g["shs"].copy())
    assert np.array_equal(inp["campos"].numpy(), g["campos"])
    return inp, g


def cov_fixture_inputs(mod):
    g = _gold("raster_k1_cov3d.npz")
    inp = util.scene_inputs(g["scales"].shape[0], 400, 400, scene_seed=41, scale_modifier=mod)
    # scene 41's activated values as the generating host computed them (torch.exp / normalize differ in the last bit between
    # CPU models, so they are taken from the fixture rather than recomputed on the testing host)
    assert np.allclose(inp["scales"].numpy(), g["scales"], rtol=1e-6) and np.allclose(inp["rotations"].numpy(), g["rotations"], atol=1e-6)
    inp["scales"] = torch.from_numpy(g["scales"].copy())
    inp["rotations"] = torch.from_numpy(g["rotations"].copy())
    return inp, g["cov3D_mod%02d" % round(mod * 10)]


def cov_err(got, want):
    """largest entry error relative to the Gaussian's trace (= sum of the squared scales).  Both the reference's torch
    code (R(q / |q|) diag(s), then L L^T as a batched matmul) and CR/forward.cu:270-304's order (M = S R, M^T M on the raw q)
    sit 1e-6 from the float64 value relative to a row's largest entry; 4.3e-7 apart on this scale."""
    return float((np.abs(got - want).max(1) / (want[:, 0] + want[:, 3] + want[:, 5])).max())


def clamp_bits(mask3):
    """the `clamped` byte of the HIP state view: bit c = channel c clamped"""
    return (mask3[:, 0].astype(np.uint8) | (mask3[:, 1].astype(np.uint8) << 1) | (mask3[:, 2].astype(np.uint8) << 2))


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_oracle_sh_to_rgb_matches_reference_eval_sh(deg):
    inp, g = sh_fixture_inputs(deg)
    fw = util.oracle_forward(inp, "FFF", with_margin=False)
    assert (fw["radii"] > 0).all()                                            # colours exist for visible Gaussians only
    want = g["rgb_deg%d" % deg]
    err = np.abs(fw["rgb"] - want).max()
    print("degree", deg, "oracle rgb vs reference eval_sh: %.2e" % err)
    assert err <= TOL
    # the clamp decision (CR/forward.cu:70-73) flips only where the raw value is within rounding of zero
    got = np.asarray(fw["clamped"]).reshape(-1, 3).astype(bool)
    differ = got != g["clamped_deg%d" % deg]
    assert (np.maximum(fw["rgb"], want)[differ] <= TOL).all() and differ.mean() < 1e-2
    assert g["clamped_deg%d" % deg].any()                                     # the fixture exercises the clamp


@pytest.mark.parametrize("mod", [1.0, 0.7])
def test_oracle_cov3d_matches_reference_build_scaling_rotation(mod):
    inp, want = cov_fixture_inputs(mod)
    fw = util.oracle_forward(inp, "FFF", with_margin=False)
    vis = fw["radii"] > 0
    assert vis.all()
    err = cov_err(fw["cov3D"], want)
    print("modifier", mod, "oracle cov3D vs reference: %.2e (relative to each Gaussian's trace)" % err)
    assert err <= TOL


@pytest.mark.parametrize("tag", ["c1", "c3", "c4"])
def test_synthetic_camera_matrices_match_reference_to_one_ulp(tag):
    g = _gold("raster_k1_cameras.npz")
    W, H = [int(v) for v in g[tag + "_size"]]
    n = g[tag + "_R"].shape[0]
    cams = S.make_cameras(n, W, H, seed=1)
    for i, c in enumerate(cams):
        assert np.array_equal(c.R, g[tag + "_R"][i]) and np.array_equal(c.T, g[tag + "_T"][i])
        for name in ("world_view_transform", "projection_matrix", "full_proj_transform", "camera_center"):
            a, b = getattr(c, name).numpy(), g[tag + "_" + name][i]
            ulp = np.spacing(np.maximum(np.abs(b), np.float32(1e-3)).astype(np.float32))
            # getWorld2View2 inverts the pose twice in float64 before rounding (utils/graphics_utils.py:106-117); ours rounds
            # the pose directly: the float32 results may differ in the last place, and campos (an inverse) by a few
            lim = 4 if name in ("camera_center", "full_proj_transform") else 1
            assert (np.abs(a - b) <= lim * ulp).all(), (tag, i, name, np.abs(a - b).max())


def _filter_cams(g):
    W, H = [int(v) for v in g["cam_size"]]
    return [S.SynthCamera(R, T, float(g["cam_fov"][0]), float(g["cam_fov"][1]), W, H) for R, T in zip(g["cam_R"], g["cam_T"])]


def test_filter3d_restatement_matches_reference_gaussian_model():
    """oracle/filter3d_ref.py (the checker of csrc/filter3d.hip) and the a7 3D-filter activation formulas against
    GaussianModel.compute_3D_filter / apply_scaling_n_opacity_with_3D_filter run on the same inputs."""
    from oracle import filter3d_ref as F
    g = _gold("raster_filter3d.npz")
    got = F.compute_3D_filter(g["xyz"], _filter_cams(g))
    assert np.abs(got - g["filter_3D"]).max() <= 2e-7 * np.abs(g["filter_3D"]).max()
    assert len(np.unique(g["filter_3D"])) > 100
    s2 = np.exp(g["log_scale"].astype(np.float64)) ** 2
    f2 = g["filter_3D"].astype(np.float64) ** 2
    want_s = np.sqrt(s2 + f2)
    want_o = 1.0 / (1.0 + np.exp(-g["opacity_logit"].astype(np.float64))) * np.sqrt(s2.prod(1) / (s2 + f2).prod(1))[:, None]
    assert np.abs(want_s - g["scales_filtered"]).max() <= 1e-6 * np.abs(want_s).max()
    assert np.abs(want_o - g["opacity_filtered"]).max() <= 1e-6
