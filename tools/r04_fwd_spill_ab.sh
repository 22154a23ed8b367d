#!/bin/bash
# Round 4 (VERDICT r3 #4a): deform_forward_b3_kernel<4,3> with 0 scratch instructions (default) vs the round-3 form with its spilled
# address pairs (-DED3_FWD_OPAQUE_ADDR=0: tools/ab_build.sh fwdspill -DED3_FWD_OPAQUE_ADDR=0), one box, three times around.
out=gpurun_out/r4g; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/e-d3dgs_amd/csrc/variants/libed3dgs_hip_fwdspill.so
ED3DGS_LIB_PATH=$V python -m pytest tests/test_deform_parity_gpu.py -q -m gpu -x -k "golden or activations_inside" > $out/pytest_variant.log 2>&1; tail -1 $out/pytest_variant.log
for rep in 1 2 3; do
  for b in noscratch spill; do
    if [ $b = spill ]; then export ED3DGS_LIB_PATH=$V; else unset ED3DGS_LIB_PATH; fi
    python bench.py --no-cpu-baseline --no-other-modes --steps 40 --warmup 10 > $out/bench_${b}_$rep.json 2> $out/bench_${b}_$rep.err
    python -c "
import json; d=json.load(open('$out/bench_${b}_$rep.json')); k=d['kernels']['deform_forward_b3_kernel<4,3>']
print('$b rep $rep ms/step %.4f median %.4f forward %.4f ms (timed region: %.4f)' % (d['ms_per_step'], d['step_ms']['median'], k['avg_launch_ms'], d['roofline']['avg_launch_ms']))"
  done
done
