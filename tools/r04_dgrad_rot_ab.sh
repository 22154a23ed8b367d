#!/bin/bash
# Round 4: the data gradient's g_a tiles with rotating weight pieces (default) vs the plain form (-DED3_DGRAD_ROT=0), one box, 3 rounds
out=gpurun_out/r4j; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_deform_parity_gpu.py tests/test_chain_parity_gpu.py -q -m gpu -x > $out/pytest.log 2>&1; tail -1 $out/pytest.log
for rep in 1 2 3; do
  for b in rot plain; do
    if [ $b = rot ]; then unset ED3DGS_LIB_PATH; else export ED3DGS_LIB_PATH=$GRAFT_REPO_ROOT/e-d3dgs_amd/csrc/variants/libed3dgs_hip_dgrot0.so; fi
    python bench.py --no-cpu-baseline --no-other-modes --steps 40 --warmup 10 > $out/bench_${b}_$rep.json 2> $out/bench_${b}_$rep.err
    python -c "
import json; d=json.load(open('$out/bench_${b}_$rep.json')); k=d['kernels']['deform_dgrad_kept_bn_kernel<4,3>']
print('$b rep $rep ms/step %.4f median %.4f dgrad %.4f ms' % (d['ms_per_step'], d['step_ms']['median'], k['avg_launch_ms']))"
  done
done
