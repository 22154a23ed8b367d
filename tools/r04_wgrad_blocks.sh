#!/bin/bash
# Round 4: block counts of the one-launch head weight gradients (ED3DGS_WG_BLOCKS_WIDE / _NARROW; default 512 / 512)
out=gpurun_out/r4u; mkdir -p $out
for cfg in "512 512" "256 512" "384 512" "512 384" "512 768" "768 512" "512 1024" "512 512"; do
  set -- $cfg
  ED3DGS_WG_BLOCKS_WIDE=$1 ED3DGS_WG_BLOCKS_NARROW=$2 python bench.py --no-cpu-baseline --no-other-modes --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d['kernels']
print('wide $1 narrow $2: ms/step %.4f median %.4f | heads %.4f all wgrad %.4f' % (d['ms_per_step'], d['step_ms']['median'], k['deform_head_wgrad_tr_all_kernel']['avg_launch_ms'], k['weight-gradient launches together']['avg_launch_ms']))"
done
