"""tools/gen_surface_golden.py -- tests/golden/raster_surface_reference.json from the REFERENCE's own Python package
`submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py` (the settings tuple, the
module, the autograd Function and what they hand to / expect from the native `_C` module).

That file is plain Python around a CUDA extension.  Here (container only: needs /root/reference) it is loaded by path with a
RECORDING stand-in registered as its `._C` (tests/support/surface_probe.py): every `_C` call's positional arguments, the order in
which the Function returns `_C`'s results, and which `_C` gradient lands in which input's .grad are written out as data.
tests/test_surface_contract_cpu.py drives this repo's package through the same probe and compares -- the binding contract of
`_C.rasterize_gaussians` / `_backward` / `mark_visible` is then the reference's own, not a transcription of it."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "support"))
import surface_probe as SP  # noqa: E402

REF_PKG = "/root/reference/submodules/diff-gaussian-rasterization/diff_gaussian_rasterization"


def main():
    import types
    holder = types.ModuleType("ref_dgr._C")
    sys.modules["ref_dgr._C"] = holder
    spec = importlib.util.spec_from_file_location("ref_dgr", os.path.join(REF_PKG, "__init__.py"), submodule_search_locations=[REF_PKG])
    pkg = importlib.util.module_from_spec(spec)
    sys.modules["ref_dgr"] = pkg
    spec.loader.exec_module(pkg)          # `from . import _C` binds the holder registered above
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_raster_golden as G
    pkg.torch = G._TorchCpu()             # integrate() builds its sub-pixel offsets with device="cuda" (:274)

    def install(rec):
        pkg._C = rec

    res = SP.probe(pkg, install)
    out = os.path.join(ROOT, "tests", "golden", "raster_surface_reference.json")
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out, "-", sum(len(v["calls"]) for v in res.values() if isinstance(v, dict) and "calls" in v), "recorded _C calls")


if __name__ == "__main__":
    main()
