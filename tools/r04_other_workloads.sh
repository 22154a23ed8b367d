#!/bin/bash
# usage (GPU box): tools/r04_other_workloads.sh  -- the bench line of the other BASELINE configs (C2, C4, C5; parity-test cases, not the headline)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r4w; mkdir -p $o
for w in C2 C4 C5; do
  timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline --no-other-modes --steps 40 --warmup 5 > $o/$w.json 2> $o/$w.err; echo "$w rc=$?"
done
python - <<PY
import json
out = {}
for w in ("C2", "C4", "C5"):
    d = json.load(open("$o/%s.json" % w))
    out[w] = {k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "step_ms", "render_fps", "config", "kernels", "roofline", "roofline_tile_backward", "roofline_tile_forward", "deform_backward_rows") if k in d}
    print(w, "ms/step %.3f  it/s %.1f  fps %.0f" % (d["ms_per_step"], d["value"], d["render_fps"]))
json.dump(out, open("$o/r04_bench_other_workloads.json", "w"), indent=1)
PY
