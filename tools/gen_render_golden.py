"""tools/gen_render_golden.py -- tests/golden/render_glue_reference.json from the REFERENCE's own gaussian_renderer/__init__.py
(render, render_tongue, render_without_tongue).

The glue is plain Python around the deformation network and the CUDA rasterizer.  Here (container only: needs /root/reference) it is
loaded by path with a RECORDING `diff_gaussian_rasterization` package, the reference's `scene.gaussian_model` / `utils.sh_utils`
behind tools/gen_raster_golden.py's import shims, `Tensor.cuda` as the identity and a `torch` proxy that drops device="cuda", and
driven by tests/support/render_probe.py: which settings it builds from a camera, what it hands the deformation network, which
activation it applies to which deformed tensor before the rasterizer sees it, how the tongue variants select rows, and the result
dictionary.  tests/test_render_glue_cpu.py drives this repo's glue through the same probe and compares.  Only data is written."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests", "support"))
import gen_raster_golden as G  # noqa: E402
import render_probe as RP  # noqa: E402


def main():
    G.load_reference_utils()
    G.load_reference_gaussian_model()                       # scene.gaussian_model (the glue imports GaussianModel for an annotation)
    sys.modules["diff_gaussian_rasterization"] = RP.make_rasterizer_package({})
    spec = importlib.util.spec_from_file_location("ref_gaussian_renderer", os.path.join(G.REF, "gaussian_renderer", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.torch = G._TorchCpu()                               # zeros_like(..., device="cuda") at :19

    def install(pkg):
        mod.GaussianRasterizationSettings, mod.GaussianRasterizer = pkg.GaussianRasterizationSettings, pkg.GaussianRasterizer

    res = RP.probe(mod, install)
    out = os.path.join(ROOT, "tests", "golden", "render_glue_reference.json")
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out, "-", len(res), "calls of the reference's glue recorded")


if __name__ == "__main__":
    main()
