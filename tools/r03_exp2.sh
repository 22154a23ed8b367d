#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3e; mkdir -p $o
python -m pytest tests/test_binning_stress_gpu.py tests/test_deform_parity_gpu.py -m gpu -q -k "handwritten or train_style or golden" > $o/pytest.log 2>&1; tail -3 $o/pytest.log
show() { python - <<PY
import json
d=json.load(open("$1")); k=d["kernels"]
print("$2", round(d["ms_per_step"],4), "med", round(d["step_ms"]["median"],4))
PY
}
python bench.py --no-cpu-baseline --no-other-modes --steps 30 --warmup 5 > $o/bench_base.json 2>/dev/null; show $o/bench_base.json base
ED3DGS_DEFORM_SIDE_STREAM=1 python bench.py --no-cpu-baseline --no-other-modes --steps 30 --warmup 5 > $o/bench_side.json 2>/dev/null; show $o/bench_side.json side_stream
for v in k89w4 k89w5; do ED3DGS_LIB_PATH=$PWD/e-d3dgs_amd/csrc/variants/libed3dgs_hip_$v.so python bench.py --no-cpu-baseline --no-other-modes --steps 30 --warmup 5 > $o/bench_$v.json 2>/dev/null; show $o/bench_$v.json $v; done
python bench.py --no-cpu-baseline --no-other-modes --steps 30 --warmup 5 > $o/bench_base2.json 2>/dev/null; show $o/bench_base2.json base_again
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $o/stats -o r --output-format csv -- python bench.py --no-cpu-baseline --no-other-modes --steps 20 --warmup 3 > $o/stats.log 2>&1
for v in k89w4 k89w5; do ED3DGS_LIB_PATH=$PWD/e-d3dgs_amd/csrc/variants/libed3dgs_hip_$v.so timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $o/stats_$v -o r --output-format csv -- python bench.py --no-cpu-baseline --no-other-modes --steps 20 --warmup 3 > $o/stats_$v.log 2>&1; done
for d in stats stats_k89w4 stats_k89w5; do f=$(ls $o/$d/*kernel_stats.csv | head -1); echo $d; grep -i "preprocess_backward" $f | cut -c1-160; done
