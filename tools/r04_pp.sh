#!/bin/bash
# Round 4: the ping-pong forward (ED3DGS_FWD_PINGPONG=1): parity of the deformation suite with it, then bench A/B on one box
# (ppbit0: teams by bit 0 of the wave index -- tools/ab_build.sh ppbit0 -DED3_FWD_PP_TEAM_BIT=0)
out=gpurun_out/r4pp; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== parity with the ping-pong forward"
ED3DGS_FWD_PINGPONG=1 timeout -k 10 400 python -m pytest tests/test_deform_parity_gpu.py tests/test_chain_parity_gpu.py -q -m gpu -x > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log
if [ $rc -ne 0 ]; then echo "parity rc=$rc: stop"; exit 1; fi
V=$GRAFT_REPO_ROOT/e-d3dgs_amd/csrc/variants/libed3dgs_hip_ppbit0.so
for rep in 1 2 3; do
  for b in default pp ppbit0; do
    unset ED3DGS_FWD_PINGPONG ED3DGS_LIB_PATH
    [ $b = pp ] && export ED3DGS_FWD_PINGPONG=1
    [ $b = ppbit0 ] && export ED3DGS_FWD_PINGPONG=1 ED3DGS_LIB_PATH=$V
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-modes --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
print('$b rep $rep: ms/step %.4f median %.4f | fwd %.4f K6 %.4f K7 %.4f | fps %.1f' % (d['ms_per_step'], d['step_ms']['median'], d['roofline']['avg_launch_ms'], d['roofline_tile_forward']['avg_launch_ms'], d['roofline_tile_backward']['avg_launch_ms'], d.get('render_fps') or 0))" || exit 1
  done
done
