"""Tile binning on adversarial inputs: the stable transpose (csrc/preprocess.hip, two-level and one-level forms) and the radix
path must give the oracle's per-tile lists, ranges and 64-bit keys BIT FOR BIT (CR/rasterizer_impl.cu:70-173, 355-395) whatever the
shape of the incidence matrix: Gaussians that cover the whole image (more than 64 super-tiles), every Gaussian piled into one
super-tile (hundreds of segments of one column), tile grids with hundreds of super-tiles, grids too wide for the two-level form,
single-tile images, and row counts around the block sizes."""
import numpy as np
import pytest
import torch

import util

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _lists_equal(inp):
    fw = util.oracle_forward(inp, "FFF", with_margin=False)
    out, sv = util.hip_forward_raw(inp, "FFF")
    assert out[0] == fw["num_rendered"]
    np.testing.assert_array_equal(out[9].cpu().numpy(), fw["radii"])
    np.testing.assert_array_equal(sv["tiles_touched"], fw["tiles_touched"])
    np.testing.assert_array_equal(sv["point_offsets"], fw["point_offsets"])
    np.testing.assert_array_equal(sv["ranges"], fw["ranges"])
    np.testing.assert_array_equal(sv["point_list"], fw["point_list"])
    np.testing.assert_array_equal(sv["keys"], fw["keys"])
    return fw


def _giants(inp, n, factor):
    inp["scales"] = inp["scales"].clone()
    inp["scales"][:n] *= factor
    return inp


CASES = {
    # 30 Gaussians 60x larger: their rects cover the whole 120 x 68 tile grid (all 135 super-tiles)
    "giants-1080p": lambda: _giants(util.scene_inputs(3000, 1920, 1080, scene_seed=5, cam_seed=6), 30, 60.0),
    # every centre inside a 0.02-wide cube: one or two super-tiles hold all rows (157 segments of one column)
    "pile": lambda: dict(util.scene_inputs(40000, 800, 608, scene_seed=7, cam_seed=8), means3D=util.scene_inputs(40000, 800, 608, scene_seed=7, cam_seed=8)["means3D"] * 0.01),
    # 240 x 135 tiles, 30 x 17 = 510 super-tiles (two threads per count column)
    "4k": lambda: _giants(util.scene_inputs(6000, 3840, 2160, scene_seed=9, cam_seed=10), 5, 40.0),
    # 257 tile columns: past the 8-bit rect of the two-level form -> the one-level transpose
    "wide-one-level": lambda: util.scene_inputs(4000, 4112, 400, scene_seed=11, cam_seed=12),
    # one tile, one super-tile
    "one-tile": lambda: util.scene_inputs(500, 16, 16, scene_seed=13, cam_seed=14),
    "one-row": lambda: util.scene_inputs(1, 400, 400, scene_seed=15, cam_seed=16),
    "1025-rows": lambda: util.scene_inputs(1025, 640, 480, scene_seed=17, cam_seed=18),
}


@pytest.mark.parametrize("case", list(CASES), ids=list(CASES))
def test_lists_bit_exact(case):
    _need_gpu()
    fw = _lists_equal(CASES[case]())
    print(case, "instances", fw["num_rendered"], "visible", int((fw["radii"] > 0).sum()))


@pytest.mark.parametrize("env", ["ED3DGS_BIN_ONE_LEVEL", "ED3DGS_BIN_RADIX"])
def test_other_binning_paths_agree(env, monkeypatch):
    """The one-level transpose and round 1's two-level radix path on a case with giants and a crowded centre."""
    _need_gpu()
    monkeypatch.setenv(env, "1")
    _lists_equal(_giants(util.scene_inputs(20000, 1100, 1604, scene_seed=19, cam_seed=20), 10, 30.0))
