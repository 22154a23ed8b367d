#!/bin/bash
# usage (GPU box): tools/r03_ab.sh <variant names...>  -- bench.py against the default library and each tools/ab_build.sh variant, twice around
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3ab; mkdir -p $o
show() { python - <<PY
import json
d=json.load(open("$1")); k=d["kernels"]
g=lambda n: round([v for kk,v in k.items() if kk.startswith(n)][0]["avg_launch_ms"],4)
print("%-12s step %.4f med %.4f | fwd %.4f dgrad %.4f wg_narrow %.4f wg_wide %.4f K6 %.4f K7 %.4f K89 %.4f | fps %d" % ("$2", d["ms_per_step"], d["step_ms"]["median"], g("deform_forward"), g("deform_dgrad"), g("deform_head_wgrad_tr_kernel<false>"), g("deform_head_wgrad_tr_kernel<true>"), g("render_forward"), g("render_backward"), g("preprocess_backward"), d["render_fps"]))
PY
}
for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-other-modes --steps 30 --warmup 5 > $o/base_$rep.json 2>/dev/null; show $o/base_$rep.json base
  for v in "$@"; do
    ED3DGS_LIB_PATH=$PWD/e-d3dgs_amd/csrc/variants/libed3dgs_hip_$v.so python bench.py --no-cpu-baseline --no-other-modes --steps 30 --warmup 5 > $o/${v}_$rep.json 2>/dev/null; show $o/${v}_$rep.json $v
  done
done
