"""The committed bench line (profiles/r02_bench_c3_final.json: one `python bench.py` run on an MI355X) carries every key
of the driver's contract, with consistent values."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_final_bench_line_has_the_contract_keys():
    d = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_c3_final.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "iters/s" and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "train iters/sec" in d["metric"] and "200k" in d["metric"] and "200k" in base["metric"]
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["gaussians"] == 200_000
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) / d["value"] < 1e-6          # whole-job throughput of one GPU
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert r["unit"] == ("TFLOP/s" if r["bound"] == "mfma" else "GB/s")
    for k in ("step_ms", "hbm_ceiling_measured_GBps", "ranks", "roofline_tile_backward"):
        assert k in d, k
    assert d["step_ms"]["p10"] <= d["step_ms"]["median"] <= d["step_ms"]["p90"] and d["step_ms"]["n"] == d["steps"]
    assert "traffic_source" in r and (r["traffic"] is None or "profiles/" in r["traffic_source"])
    v = d["roofline_tile_backward"]["valu_roof"]
    assert 0 < v["frac"] < 1 and v["blended_pairs_per_launch"] > v["visited_iterations_per_launch"] > 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == d["unit"] and 0 < c["value"] < d["value"]
