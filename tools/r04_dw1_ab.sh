#!/bin/bash
# Round 4: dW1 stream launch with 512 (default) / 768 / 1024 blocks
out=gpurun_out/r4p; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for b in 512 256 384; do
    if [ $b = 512 ]; then unset ED3DGS_LIB_PATH; else export ED3DGS_LIB_PATH=$GRAFT_REPO_ROOT/e-d3dgs_amd/csrc/variants/libed3dgs_hip_dw1b$b.so; fi
    python bench.py --no-cpu-baseline --no-other-modes --steps 40 --warmup 10 > $out/bench_${b}_$rep.json 2> $out/bench_${b}_$rep.err
    python -c "
import json; d=json.load(open('$out/bench_${b}_$rep.json'))
print('$b rep $rep ms/step %.4f median %.4f dw1 %.4f ms' % (d['ms_per_step'], d['step_ms']['median'], d['kernels']['deform_dw1_kernel']['avg_launch_ms']))"
  done
done
