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
    # _C.integrate_gaussians_to_points: 23 positional arguments.  One documented difference: argument 15, `subpixel_offset`, is a
    # freshly filled (H, W, 2) zero tensor in the reference (:274) and None here -- neither native side reads it
    # (DGR/rasterize_points.cu / diff_gaussian_rasterization/_C.py: accepted and ignored), so this surface does not fill 8 H W bytes per call
    rc, gc = ref["integrate"]["calls"][0], got["integrate"]["calls"][0]
    assert rc[0] == gc[0] == "integrate_gaussians_to_points" and len(rc[1]) == len(gc[1]) == 23
    assert rc[1][15] == ["tensor", 0.0, [4, 6, 2]] and gc[1][15] == ["NoneType", "None"]
    assert [a for i, a in enumerate(gc[1]) if i != 15] == [a for i, a in enumerate(rc[1]) if i != 15]
    assert got["integrate"]["returned"] == ref["integrate"]["returned"]
    assert got["argument_errors"] == ref["argument_errors"]
