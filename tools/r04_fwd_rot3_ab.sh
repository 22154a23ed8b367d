#!/bin/bash
# Round 4: head tiles of the forward with the weight pieces of a WHOLE k-step requested one step ahead (two register sets, the narrow
# heads' bias loads moved behind the products): -DED3_FWD_ROT=3 (tools/ab_build.sh fwdrot3 -DED3_FWD_ROT=3) against the default (2)
out=gpurun_out/r4rot3; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/e-d3dgs_amd/csrc/variants/libed3dgs_hip_fwdrot3.so
echo "== parity with the variant"
ED3DGS_LIB_PATH=$V timeout -k 10 400 python -m pytest tests/test_deform_parity_gpu.py tests/test_chain_parity_gpu.py -q -m gpu -x > $out/pytest.log 2>&1; rc=$?; tail -2 $out/pytest.log
[ $rc -ne 0 ] && exit 1
for rep in 1 2 3; do
  for b in default rot3; do
    unset ED3DGS_LIB_PATH; [ $b = rot3 ] && export ED3DGS_LIB_PATH=$V
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-modes --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
print('$b rep $rep: ms/step %.4f median %.4f | fwd %.4f | fps %.1f' % (d['ms_per_step'], d['step_ms']['median'], d['roofline']['avg_launch_ms'], d.get('render_fps') or 0))"
  done
done
