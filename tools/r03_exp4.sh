#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3g; mkdir -p $o
python -m pytest tests -m gpu -q > $o/pytest.log 2>&1; tail -6 $o/pytest.log
show() { python - <<PY
import json
d=json.load(open("$1")); k=d["kernels"]
print("$2", round(d["ms_per_step"],4), "med", round(d["step_ms"]["median"],4), "K6", round(k["render_forward_kernel<false,true> (K6)"]["avg_launch_ms"],4), "K7", round(k["render_backward_kernel<false,true> (K7)"]["avg_launch_ms"],4), "fps", round(d["render_fps"]))
PY
}
python bench.py --no-cpu-baseline --no-other-modes --steps 30 --warmup 5 > $o/bench_base.json 2>$o/bench_base.err; show $o/bench_base.json quadrant_sched
ED3DGS_LIB_PATH=$PWD/e-d3dgs_amd/csrc/variants/libed3dgs_hip_k6w4.so python bench.py --no-cpu-baseline --no-other-modes --steps 30 --warmup 5 > $o/bench_k6w4.json 2>/dev/null; show $o/bench_k6w4.json k6w4
ED3DGS_LIB_PATH=$PWD/e-d3dgs_amd/csrc/variants/libed3dgs_hip_k6w6.so python bench.py --no-cpu-baseline --no-other-modes --steps 30 --warmup 5 > $o/bench_k6w6.json 2>/dev/null; show $o/bench_k6w6.json k6w6
