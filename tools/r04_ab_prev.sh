#!/bin/bash
# usage (GPU box, repo root): tools/r04_ab_prev.sh <out-name> [pytest files...]   -- parity of the default build, then three rounds of
# bench A/B against e-d3dgs_amd/csrc/variants/libed3dgs_hip_prev.so (the previous commit's library, built in a worktree)
name=$1; shift
out=gpurun_out/$name; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tests="$*"; [ -z "$tests" ] && tests="tests/test_raster_parity_gpu.py tests/test_fullsize_gpu.py tests/test_odd_sizes_gpu.py tests/test_reference_paths_gpu.py tests/test_render_variants_gpu.py tests/test_chain_parity_gpu.py"
echo "== parity"
timeout -k 10 700 python -m pytest $tests -q -m gpu -x > $out/pytest.log 2>&1; tail -3 $out/pytest.log
V=$GRAFT_REPO_ROOT/e-d3dgs_amd/csrc/variants/libed3dgs_hip_prev.so
for rep in 1 2 3; do
  for b in default prev; do
    if [ $b = prev ]; then export ED3DGS_LIB_PATH=$V; else unset ED3DGS_LIB_PATH; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-modes --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
k=d['kernels']
print('$b rep $rep: ms/step %.4f median %.4f | fwd %.4f K6 %.4f K7 %.4f K8+K9 %.4f | fps %.1f' % (d['ms_per_step'], d['step_ms']['median'], d['roofline']['avg_launch_ms'] if 'avg_launch_ms' in d['roofline'] else -1, d['roofline_tile_forward']['avg_launch_ms'], d['roofline_tile_backward']['avg_launch_ms'], k['preprocess_backward_kernel (K8+K9)']['avg_launch_ms'], d.get('render_fps') or 0))"
  done
done
