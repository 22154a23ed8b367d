"""SURVEY 8 rows a1-a2, pinned by the reference's own Python: this repo's `gaussian_renderer` glue (render, render_tongue,
render_without_tongue) driven through tests/support/render_probe.py -- fake camera, fake model with MARKED activations, recording
deformation network, recording rasterizer -- against tests/golden/render_glue_reference.json, the SAME probe run on the reference's
gaussian_renderer/__init__.py (tools/gen_render_golden.py, container only).  Compared per call: every raster setting (as the number
the native side receives), the nine positional arguments and the keywords handed to the deformation network, every keyword handed to
the rasterizer -- i.e. which activation was applied to which deformed tensor, with and without the 3D filter, and which rows the
tongue variants select -- and the whole result dictionary.  The two documented differences are asserted as such: the camera time is a
number here and a (P, 1) tensor there (same value), and with `override_color` the reference passes BOTH `shs` and `colors_precomp`,
which its own rasterizer rejects (DGR/diff_gaussian_rasterization/__init__.py:213); this glue passes the colours only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "e-d3dgs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "support"))


def _ours():
    import render_probe as RP
    import gaussian_renderer as mod
    saved = (mod.GaussianRasterizationSettings, mod.GaussianRasterizer)

    def install(pkg):
        mod.GaussianRasterizationSettings, mod.GaussianRasterizer = pkg.GaussianRasterizationSettings, pkg.GaussianRasterizer
    try:
        return RP.probe(mod, install)
    finally:
        mod.GaussianRasterizationSettings, mod.GaussianRasterizer = saved


def test_render_glue_does_what_the_references_does():
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "render_glue_reference.json")))
    got = json.loads(json.dumps(_ours()))
    assert [(c["function"], c["kwargs"]) for c in got] == [(c["function"], c["kwargs"]) for c in ref] and len(ref) == 8
    for r, g in zip(ref, got):
        tag = (r["function"], r["kwargs"])
        assert g["settings"] == r["settings"], tag
        assert g["deformation_args"] == r["deformation_args"], tag           # (the time as its value: tensor there, number here)
        assert g["deformation_kwargs"] == r["deformation_kwargs"], tag       # iter, num_down_emb_c, num_down_emb_f
        if r["function"] == "integrate":   # rasterizer.integrate's keywords: filtered scales + opacity, activated rotations, SH as they come
            assert g["integrate_kwargs"] == r["integrate_kwargs"], tag
            assert g["result"] == r["result"], tag
            continue
        rk, gk = dict(r["rasterizer_kwargs"]), dict(g["rasterizer_kwargs"])
        if "override_color" in r["kwargs"]:
            assert rk["shs"] is not None and rk["colors_precomp"] is not None   # the reference hands over both (and its rasterizer raises)
            assert gk["shs"] is None                                            # here: the colours only
            rk["shs"] = None
        assert gk == rk, tag
        assert g["result"] == r["result"], tag
    # the probe saw what it was built to see: marked activations on the deformed tensors, the filter variant, the tongue rows
    by = {(c["function"], tuple(c["kwargs"])): c for c in ref}
    plain = by[("render", ())]["rasterizer_kwargs"]
    assert plain["scales"][1] == 44.0 and plain["rotations"][1] == 69.0 and plain["opacities"][1] == 120.0
    filt = by[("render", ("disable_filter3D", "require_coord", "scaling_modifier"))]["rasterizer_kwargs"]
    assert filt["scales"][1] == 154.0 and filt["opacities"][1] == 264.0
    assert by[("render_tongue", ())]["rasterizer_kwargs"]["tongue_class"][1] == [0.7, 1.0, 0.6]
    assert by[("render_without_tongue", ())]["rasterizer_kwargs"]["tongue_class"][1] == [0.2, 0.0]
