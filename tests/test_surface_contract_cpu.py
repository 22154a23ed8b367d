"""SURVEY 8 rows a8-a11, pinned by the reference's own Python: this repo's `diff_gaussian_rasterization` package driven through
tests/support/surface_probe.py with a recording stand-in for its native `_C`, against tests/golden/raster_surface_reference.json --
the SAME probe run on the reference's `diff_gaussian_rasterization/__init__.py` (tools/gen_surface_golden.py, container only).
Compared: the settings tuple's fields, every positional argument of `_C.rasterize_gaussians`, `_C.rasterize_gaussians_backward` and
`_C.mark_visible` (for the SH + scales / rotations call and for the precomputed colour + covariance call), the order in which the
Function returns `_C`'s results, which `_C` gradient lands in which input's .grad (None for tongue_class), and the argument-check
error texts.  No GPU and no native code: the contract of the boundary, not the kernels behind it."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "e-d3dgs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "support"))


def _ours():
    import surface_probe as SP
    import diff_gaussian_rasterization as pkg
    saved = pkg._C

    def install(rec):
        pkg._C = rec
    try:
        return SP.probe(pkg, install)
    finally:
        pkg._C = saved


def test_the_python_surface_hands_C_what_the_references_does():
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "raster_surface_reference.json")))
    got = json.loads(json.dumps(_ours()))          # (tuples -> lists, as in the fixture)
    assert got["settings_fields"] == ref["settings_fields"] and len(ref["settings_fields"]) == 15
    for variant in ("sh_scales_rotations", "precomputed_colour_and_covariance"):
        r, g = ref[variant], got[variant]
        assert [c[0] for c in g["calls"]] == [c[0] for c in r["calls"]] == ["rasterize_gaussians", "rasterize_gaussians_backward"]
        for (name, rargs), (_, gargs) in zip(r["calls"], g["calls"]):
            assert len(gargs) == len(rargs), name
            for i, (ra, ga) in enumerate(zip(rargs, gargs)):
                assert ga == ra, (variant, name, i, ga, ra)
        assert g["returned"] == r["returned"], variant            # (color, radii, coord, mcoord, depth, mdepth, alpha, tongue, normal)
        assert g["input_grads"] == r["input_grads"], variant      # which _C gradient lands where; tongue_class gets none
    assert len(ref["sh_scales_rotations"]["calls"][0][1]) == 23 and len(ref["sh_scales_rotations"]["calls"][1][1]) == 32
    assert got["mark_visible"] == ref["mark_visible"]
    assert got["argument_errors"] == ref["argument_errors"]
