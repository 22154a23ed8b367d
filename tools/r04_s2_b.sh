#!/bin/bash
# Round 4, session 2, call B: K6 with the done flags inside the per-pixel alpha threshold (parity + A/B against the previous K6),
# the dist module through RCCL (one-rank communicator), bench line with its collective on RCCL
out=gpurun_out/r4s2b; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== parity"
timeout -k 10 600 python -m pytest tests/test_raster_parity_gpu.py tests/test_fullsize_gpu.py tests/test_odd_sizes_gpu.py tests/test_reference_paths_gpu.py tests/test_render_variants_gpu.py tests/test_chain_parity_gpu.py tests/test_dist_rccl_gpu.py -q -m gpu -x > $out/pytest.log 2>&1; tail -3 $out/pytest.log
V=$GRAFT_REPO_ROOT/e-d3dgs_amd/csrc/variants/libed3dgs_hip_k6old.so
for rep in 1 2 3; do
  for b in default old; do
    if [ $b = old ]; then export ED3DGS_LIB_PATH=$V; else unset ED3DGS_LIB_PATH; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-modes --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
k=d['kernels']
print('$b rep $rep: ms/step %.4f median %.4f | K6 %.4f K7 %.4f ms | fps %s' % (d['ms_per_step'], d['step_ms']['median'], d['roofline_tile_forward']['avg_launch_ms'] if 'roofline_tile_forward' in d else -1, d['roofline_tile_backward']['avg_launch_ms'], d.get('render_fps')))"
  done
done
unset ED3DGS_LIB_PATH
echo "== bench with its collective on RCCL (one-rank communicator)"
ED3DGS_DIST_COLLECTIVES_AT_WORLD_1=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-modes --steps 20 --warmup 5 > $out/bench_rccl_world1.json 2> $out/bench_rccl_world1.err; echo rc=$?; python -c "
import json; d=json.load(open('$out/bench_rccl_world1.json')); print('ms/step', d['ms_per_step'], 'ranks', d.get('ranks'))"
ED3DGS_DIST_COLLECTIVES_AT_WORLD_1=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-modes --steps 20 --warmup 5 --dp-grads > $out/bench_rccl_world1_dp.json 2> $out/bench_rccl_world1_dp.err; echo rc=$?; python -c "
import json; d=json.load(open('$out/bench_rccl_world1_dp.json')); print('dp-grads ms/step', d['ms_per_step'], 'ranks', d.get('ranks'))"
tail -3 $out/bench_rccl_world1_dp.err
