#!/bin/bash
# Round 4: the head weight-gradient kernels' dW2 products with rotating B pieces (default) vs the plain form (-DED3_WGRAD_ROT=0)
out=gpurun_out/r4k; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_deform_parity_gpu.py tests/test_chain_parity_gpu.py -q -m gpu -x > $out/pytest.log 2>&1; tail -1 $out/pytest.log
for rep in 1 2 3; do
  for b in rot plain; do
    if [ $b = rot ]; then unset ED3DGS_LIB_PATH; else export ED3DGS_LIB_PATH=$GRAFT_REPO_ROOT/e-d3dgs_amd/csrc/variants/libed3dgs_hip_wgrot0.so; fi
    python bench.py --no-cpu-baseline --no-other-modes --steps 40 --warmup 10 > $out/bench_${b}_$rep.json 2> $out/bench_${b}_$rep.err
    python -c "
import json; d=json.load(open('$out/bench_${b}_$rep.json')); k=d['kernels']
print('$b rep $rep ms/step %.4f median %.4f narrow %.4f wide %.4f ms' % (d['ms_per_step'], d['step_ms']['median'], k['deform_head_wgrad_tr_kernel<false>']['avg_launch_ms'], k['deform_head_wgrad_tr_kernel<true>']['avg_launch_ms']))"
  done
done
