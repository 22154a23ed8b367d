#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3h; mkdir -p $o
python -m pytest tests/test_raster_parity_gpu.py tests/test_render_variants_gpu.py tests/test_odd_sizes_gpu.py tests/test_chain_parity_gpu.py tests/test_fullsize_gpu.py tests/test_integrate_gpu.py -m gpu -q > $o/pytest.log 2>&1; tail -4 $o/pytest.log
python bench.py --no-cpu-baseline --no-other-modes --steps 30 --warmup 5 > $o/bench_base.json 2>$o/bench_base.err
python - <<PY
import json
d=json.load(open("$o/bench_base.json")); k=d["kernels"]
print("prefetch", round(d["ms_per_step"],4), "med", round(d["step_ms"]["median"],4), "K6", round(k["render_forward_kernel<false,true> (K6)"]["avg_launch_ms"],4), "K7", round(k["render_backward_kernel<false,true> (K7)"]["avg_launch_ms"],4), "fps", round(d["render_fps"]))
PY
