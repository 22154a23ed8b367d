#!/bin/bash
# Round 4: how the driver's command (20 steps, 5 warm-up) spreads from run to run on one box: windows of 4 steps, wall per step
out=gpurun_out/r4out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5 6 7 8; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-modes --mark-every 1 2>/dev/null > $out/run_$rep.json
  python -c "
import json; d=json.load(open('$out/run_$rep.json'))
w=d['step_ms_windows']
print('rep $rep wall/step %.4f  gpu mean %.4f median %.4f max %.3f | %s | mallocs %s' % (d['ms_per_step'], sum(w)/len(w), d['step_ms']['median'], max(w), ' '.join('%.2f'%x for x in w), d.get('device_mallocs_in_timed_region')))"
done
