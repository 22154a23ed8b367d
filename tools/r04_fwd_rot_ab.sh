#!/bin/bash
# Round 4: the forward's head-tile MFMA loop -- plain (ED3_FWD_ROT=0), weight pieces rotating through their registers with the
# compiler's waits (=1: lgkmcnt(0) after any LDS-DMA), rotating with counted waits in inline assembly (=2, default).  One box, 3 rounds.
out=gpurun_out/r4i; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_deform_parity_gpu.py -q -m gpu -x > $out/pytest_rot2.log 2>&1; tail -1 $out/pytest_rot2.log
for rep in 1 2 3; do
  for b in rot2 rot0 rot1; do
    if [ $b = rot2 ]; then unset ED3DGS_LIB_PATH; else export ED3DGS_LIB_PATH=$GRAFT_REPO_ROOT/e-d3dgs_amd/csrc/variants/libed3dgs_hip_$b.so; fi
    python bench.py --no-cpu-baseline --no-other-modes --steps 40 --warmup 10 > $out/bench_${b}_$rep.json 2> $out/bench_${b}_$rep.err
    python -c "
import json; d=json.load(open('$out/bench_${b}_$rep.json')); k=d['kernels']['deform_forward_b3_kernel<4,3>']
print('$b rep $rep ms/step %.4f median %.4f forward %.4f ms (timed region: %.4f) fps %.1f' % (d['ms_per_step'], d['step_ms']['median'], k['avg_launch_ms'], d['roofline']['avg_launch_ms'], d['render_fps']))"
  done
done
