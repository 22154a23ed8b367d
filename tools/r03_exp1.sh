#!/bin/bash
# round-3 experiment batch 1 (one gpurun call): sort timing, six-product errors / speed, block stagger
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3d; mkdir -p $o
python -m pytest tests/test_binning_stress_gpu.py tests/test_deform_parity_gpu.py -m gpu -q -k "binning or lists or train_style or library" > $o/pytest.log 2>&1; tail -3 $o/pytest.log
timeout -k 10 300 rocprofv3 --kernel-trace -d $o/trace -o r --output-format csv -- python bench.py --no-cpu-baseline --no-other-modes --steps 6 --warmup 2 > $o/trace.log 2>&1; python tools/step_trace.py $o/trace > $o/step_trace.txt 2>&1; grep -i "sort\|launches" $o/step_trace.txt
python tools/six_product_errors.py eight > $o/err_eight.json 2> $o/err_eight.err
ED3DGS_DEFORM_FP32_MFMA=1 python tools/six_product_errors.py fp32_mfma > $o/err_fp32.json 2> $o/err_fp32.err
ED3DGS_LIB_PATH=$PWD/e-d3dgs_amd/csrc/variants/libed3dgs_hip_six.so python tools/six_product_errors.py six > $o/err_six.json 2> $o/err_six.err
for f in eight fp32 six; do python - <<PY
import json
d=json.load(open("$o/err_$f.json")); print(d["label"], d["worst"], "ms", round(d["ms_fwd_bwd_65836"],3))
PY
done
for st in 0 20 40 80; do ED3DGS_DEFORM_STAGGER=$st python bench.py --no-cpu-baseline --no-other-modes --steps 30 --warmup 5 > $o/bench_stagger$st.json 2>/dev/null; python - <<PY
import json
d=json.load(open("$o/bench_stagger$st.json")); k=d["kernels"]
print("stagger $st", round(d["ms_per_step"],4), "fwd", round(k["deform_forward_b3_kernel<4,3>"]["avg_launch_ms"],4), "dgrad", round(k["deform_dgrad_kept_bn_kernel<4,3>"]["avg_launch_ms"],4))
PY
done
ED3DGS_LIB_PATH=$PWD/e-d3dgs_amd/csrc/variants/libed3dgs_hip_six.so python bench.py --no-cpu-baseline --no-other-modes --steps 30 --warmup 5 > $o/bench_six.json 2>/dev/null; python - <<PY
import json
d=json.load(open("$o/bench_six.json")); k=d["kernels"]
print("six", round(d["ms_per_step"],4), {n[:28]: round(v["avg_launch_ms"],4) for n,v in k.items() if isinstance(v,dict) and "deform" in n})
PY
