#!/bin/bash
# Round 4: how the driver's command (20 steps, 5 warm-up) spreads from run to run on one box: per-step GPU marks, the host's time per step
# split into render() (which waits for K1's count) and backward().  usage: tools/r04_outliers.sh [runs]   (default 8)
out=gpurun_out/r4out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
n=${1:-8}
for rep in $(seq 1 $n); do
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-modes --mark-every 1 2>/dev/null > $out/run_$rep.json
  python -c "
import json; d=json.load(open('$out/run_$rep.json'))
w=d['step_ms_windows']; hs=d['host_ms_in_render_and_backward']
k=max(range(len(w)), key=lambda i: w[i])
print('rep $rep wall/step %.4f  gpu mean %.4f median %.4f max %.3f at step %d (host render %.3f backward %.3f; medians %.3f / %.3f)' % (d['ms_per_step'], sum(w)/len(w), d['step_ms']['median'], w[k], k, hs[k][0], hs[k][1], sorted(h[0] for h in hs)[len(hs)//2], sorted(h[1] for h in hs)[len(hs)//2]))"
done
