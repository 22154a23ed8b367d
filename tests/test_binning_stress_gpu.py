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


def _lists_equal(inp, path=None):
    """`path`: the binning back end the frame must take (ed3dgs_binning_path: 2 two-level transpose, 1 one-level, 0 radix)."""
    if path is not None:
        from ed3dgs_amd import _lib
        assert _lib.lib().ed3dgs_binning_path(int(inp["P"]), int(inp["W"]), int(inp["H"])) == path
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


# case -> (inputs, the back end it must take)
CASES = {
    # 30 Gaussians 60x larger: their rects cover the whole 120 x 68 tile grid (all 135 super-tiles)
    "giants-1080p": lambda: (_giants(util.scene_inputs(3000, 1920, 1080, scene_seed=5, cam_seed=6), 30, 60.0), 2),
    # every centre inside a 0.02-wide cube: one or two super-tiles hold all rows (157 segments of one column)
    "pile": lambda: (dict(util.scene_inputs(40000, 800, 608, scene_seed=7, cam_seed=8), means3D=util.scene_inputs(40000, 800, 608, scene_seed=7, cam_seed=8)["means3D"] * 0.01), 2),
    # 240 x 135 tiles = 32400 (more than any LDS counter array of the one-level form), 30 x 17 = 510 super-tiles: two threads
    # per count column in level A's column sums, 8 blocks of 64 super-tiles per row mask
    "4k": lambda: (_giants(util.scene_inputs(6000, 3840, 2160, scene_seed=9, cam_seed=10), 5, 40.0), 2),
    # the same grid with enough rows for several level-A blocks and giants that cover all 510 super-tiles
    "4k-giants-9k": lambda: (_giants(util.scene_inputs(9000, 3840, 2160, scene_seed=21, cam_seed=22), 12, 80.0), 2),
    # 128 x 96 tiles = 12288 (the one-level form's limit exactly), 16 x 12 = 192 super-tiles: 5 threads per column, 3 mask blocks
    "2048x1536": lambda: (_giants(util.scene_inputs(5000, 2048, 1536, scene_seed=23, cam_seed=24), 8, 50.0), 2),
    # 250 x 250 tiles, 32 x 32 = 1024 super-tiles: the two-level form's limit (one thread per column, 16 mask blocks, 74 KB of LDS)
    "4000x4000": lambda: (_giants(util.scene_inputs(3000, 4000, 4000, scene_seed=25, cam_seed=26), 4, 60.0), 2),
    # 257 tile columns: past the 8-bit rect of the two-level form -> the one-level transpose
    "wide-one-level": lambda: (util.scene_inputs(4000, 4112, 400, scene_seed=11, cam_seed=12), 1),
    # 257 x 60 tiles = 15420 > 12288: neither transpose form -> round 1's scan + K3 + radix sort + K5
    "wide-radix": lambda: (util.scene_inputs(4000, 4112, 960, scene_seed=27, cam_seed=28), 0),
    # one tile, one super-tile
    "one-tile": lambda: (util.scene_inputs(500, 16, 16, scene_seed=13, cam_seed=14), 2),
    "one-row": lambda: (util.scene_inputs(1, 400, 400, scene_seed=15, cam_seed=16), 2),
    "1025-rows": lambda: (util.scene_inputs(1025, 640, 480, scene_seed=17, cam_seed=18), 2),
    # every Gaussian at the same point: ONE depth key, every tie resolved by id (the depth sort's digits are 1 bit wide)
    "identical-depths": lambda: (dict(util.scene_inputs(3000, 640, 480, scene_seed=29, cam_seed=30),
                                      means3D=util.scene_inputs(3000, 640, 480, scene_seed=29, cam_seed=30)["means3D"] * 0.0), 2),
    # depth keys over many binades (the depth sort's widest digits) and more than one sort tile of 4096 keys
    "deep-range-9k": lambda: (dict(util.scene_inputs(9000, 800, 608, scene_seed=31, cam_seed=32),
                                   means3D=util.scene_inputs(9000, 800, 608, scene_seed=31, cam_seed=32)["means3D"]
                                   * torch.logspace(-2.5, 1.0, 9000)[:, None]), 2),
}


@pytest.mark.parametrize("case", list(CASES), ids=list(CASES))
def test_lists_bit_exact(case):
    _need_gpu()
    fw = _lists_equal(*CASES[case]())
    print(case, "instances", fw["num_rendered"], "visible", int((fw["radii"] > 0).sum()))


@pytest.mark.parametrize("case", ["giants-1080p", "pile", "identical-depths", "deep-range-9k", "one-row", "1025-rows"])
def test_handwritten_depth_sort_agrees(case, libopt):
    """ED3DGS_SORT_HANDWRITTEN=1: the bucket + rank sort of csrc/binning.hip (round 4: 3 launches) in place of the library's stable
    sort for binning level 1 -- opt-in: bit-identical, fewer launches, slower on average."""
    _need_gpu()
    libopt("SORT_HANDWRITTEN", 1)
    _lists_equal(*CASES[case]())


def test_depth_order_is_the_stable_sort_of_the_keys():
    """The whole permutation, not only what the tile lists show of it: order[] of the hand-written sort against torch's stable
    sort of the depth keys, culled Gaussians (key 0xFFFFFFFF) last and in id order, on a frame with ties, culled rows, a deep
    depth range and more than one bucket block; and against the library's sort."""
    _need_gpu()
    from ed3dgs_amd import _lib
    inp = dict(util.scene_inputs(30000, 800, 608, scene_seed=41, cam_seed=42))
    m = inp["means3D"].clone()
    m[::7] = m[3]                                         # thousands of exact ties
    m[1::5] *= torch.logspace(-2.0, 1.2, m[1::5].shape[0])[:, None]   # depths over many binades, some behind the camera
    inp["means3D"] = m
    orders = {}
    for lib in (0, 1):
        old = _lib.set_option("SORT_HANDWRITTEN", 1 - lib)
        try:
            out, sv = util.hip_forward_raw(inp, "FFF")
            orders[lib] = sv["depth_order"].astype(np.int64)
        finally:
            _lib.set_option("SORT_HANDWRITTEN", old)
        keys = np.where(out[9].cpu().numpy() > 0, sv["depths"].view(np.uint32).astype(np.int64), 0xFFFFFFFF)
    vis = out[9].cpu().numpy() > 0
    assert 0.05 < vis.mean() < 0.99
    # K1's key of a culled Gaussian is 0xFFFFFFFF whatever its depth; a visible one's is its depth bits
    want = torch.sort(torch.from_numpy(keys), stable=True).indices.numpy()
    assert np.array_equal(orders[0], want)
    assert np.array_equal(orders[1], want)


@pytest.mark.parametrize("switch,path", [("BIN_ONE_LEVEL", 1), ("BIN_RADIX", 0)])
def test_other_binning_paths_agree(switch, path, libopt):
    """The one-level transpose and round 1's two-level radix path on a case with giants and a crowded centre."""
    _need_gpu()
    libopt(switch, 1)
    _lists_equal(_giants(util.scene_inputs(20000, 1100, 1604, scene_seed=19, cam_seed=20), 10, 30.0), path)


@pytest.mark.parametrize("case", ["giants-1080p", "one-row", "1025-rows", "wide-radix"])
def test_count_by_copy_agrees(case, libopt):
    """The instance count reaches the host through a host-coherent word K1's last block writes (csrc/common.h, CountMail);
    ED3DGS_COUNT_COPY=1 is the round-2 form (per-block sums copied to pinned memory behind an event).  Same count, same lists --
    and the default form again afterwards (its sequence number and counter survive a call of the other form)."""
    _need_gpu()
    ref = _lists_equal(*CASES[case]())
    libopt("COUNT_COPY", 1)
    got = _lists_equal(*CASES[case]())
    assert got["num_rendered"] == ref["num_rendered"]
    libopt("COUNT_COPY", 0)
    again = _lists_equal(*CASES[case]())
    assert again["num_rendered"] == ref["num_rendered"]
