#!/bin/bash
# usage (GPU box): tools/r03_env_ab.sh NAME=VALUE ...  -- bench.py with the default switches and with each environment setting, twice around
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3ab; mkdir -p $o
show() { python - <<PY
import json
d=json.load(open("$1"))
print("%-22s step %.4f med %.4f | fps %d" % ("$2", d["ms_per_step"], d["step_ms"]["median"], d["render_fps"]))
PY
}
for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-other-modes --steps 40 --warmup 5 > $o/envbase_$rep.json 2>/dev/null; show $o/envbase_$rep.json default
  for v in "$@"; do
    env "$v" python bench.py --no-cpu-baseline --no-other-modes --steps 40 --warmup 5 > $o/env_${v%%=*}_$rep.json 2>/dev/null; show $o/env_${v%%=*}_$rep.json "$v"
  done
done
