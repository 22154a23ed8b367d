#!/bin/bash
# usage: tools/ab.sh VAR [steps]  -- same box, alternating runs with and without VAR=1: median step of each
steps=${2:-30}
for rep in 1 2 3; do
  for v in 0 1; do
    if [ $v = 1 ]; then export $1=1; else unset $1; fi
    python bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-other-modes 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$1=$v', 'median %.4f' % d['step_ms']['median'], 'mean %.4f' % d['ms_per_step'], 'fps %.1f' % d['render_fps'])"
  done
done
