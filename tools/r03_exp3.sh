#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3f; mkdir -p $o
python -m pytest tests -m gpu -q > $o/pytest.log 2>&1; tail -6 $o/pytest.log
show() { python - <<PY
import json
d=json.load(open("$1")); k=d["kernels"]
print("$2", round(d["ms_per_step"],4), "med", round(d["step_ms"]["median"],4), "K6", round(k["render_forward_kernel<false,true> (K6)"]["avg_launch_ms"],4), "K7", round(k["render_backward_kernel<false,true> (K7)"]["avg_launch_ms"],4), "fps", round(d["render_fps"]))
v=d["roofline_tile_backward"]["valu_roof"]; print("  K7 iters", v["visited_iterations_per_launch"], "pairs/iter", round(v["pairs_per_iteration"],1), "qentries/iter", round(v.get("quadrant_entries_per_iteration",0),2))
v=d["roofline_tile_forward"]["valu_roof"]; print("  K6 iters", v["visited_iterations_per_launch"], "pairs/iter", round(v["pairs_per_iteration"],1))
PY
}
python bench.py --no-cpu-baseline --no-other-modes --steps 30 --warmup 5 > $o/bench_base.json 2>$o/bench_base.err; show $o/bench_base.json quadrant_sched
