#!/bin/bash
# Round 4: Gaussians per block of the per-Gaussian backward (K8+K9; -DED3_K89_ROWS, default 512)
out=gpurun_out/r4v; mkdir -p $out
for rep in 1 2; do
for v in default k89r256 k89r768 k89r1024; do
  if [ $v = default ]; then unset ED3DGS_LIB_PATH; else export ED3DGS_LIB_PATH=$GRAFT_REPO_ROOT/e-d3dgs_amd/csrc/variants/libed3dgs_hip_$v.so; fi
  python bench.py --no-cpu-baseline --no-other-modes --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d['kernels']
print('$v rep $rep: ms/step %.4f median %.4f | K8+K9 %.4f ms' % (d['ms_per_step'], d['step_ms']['median'], k['preprocess_backward_kernel (K8+K9)']['avg_launch_ms']))"
done; done
