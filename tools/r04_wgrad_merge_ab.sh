#!/bin/bash
# Round 4: the SH head's and the narrow heads' weight gradients as ONE launch (default) vs two (ED3DGS_WGRAD_SEPARATE=1)
out=gpurun_out/r4t; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_deform_parity_gpu.py tests/test_chain_parity_gpu.py -q -m gpu -x > $out/pytest.log 2>&1; tail -1 $out/pytest.log
for rep in 1 2 3; do
  for b in merged separate; do
    if [ $b = merged ]; then unset ED3DGS_WGRAD_SEPARATE; else export ED3DGS_WGRAD_SEPARATE=1; fi
    python bench.py --no-cpu-baseline --no-other-modes --steps 40 --warmup 10 > $out/bench_${b}_$rep.json 2> $out/bench_${b}_$rep.err
    python -c "
import json; d=json.load(open('$out/bench_${b}_$rep.json')); k=d['kernels']
print('$b rep $rep ms/step %.4f median %.4f | all weight-gradient launches %.4f ms | ' % (d['ms_per_step'], d['step_ms']['median'], k['weight-gradient launches together']['avg_launch_ms']) + ' '.join('%s %.4f' % (n.split('_kernel')[0][-14:], v['avg_launch_ms']) for n, v in k.items() if 'wgrad' in n or 'dw1' in n))"
  done
done
