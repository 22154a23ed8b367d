#!/bin/bash
# Round 4, K7 gradient-record experiment in one GPU-box call: parity of the LDS-tile build, bench A/B, and
# FETCH_SIZE / WRITE_SIZE passes (every launch a training-step launch) for both builds.
# usage (GPU box, repo root): tools/r04_k7_ab.sh   (needs e-d3dgs_amd/csrc/variants/libed3dgs_hip_k7tile.so: tools/ab_build.sh k7tile -DED3_K7_LDS_TILE=1)
out=gpurun_out/r4k7; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/e-d3dgs_amd/csrc/variants/libed3dgs_hip_k7tile.so
echo "== parity, LDS-tile build"; ED3DGS_LIB_PATH=$V python -m pytest tests/test_raster_parity_gpu.py tests/test_fullsize_gpu.py tests/test_odd_sizes_gpu.py tests/test_reference_paths_gpu.py -q -m gpu -x > $out/pytest_tile.log 2>&1; tail -2 $out/pytest_tile.log
for rep in 1 2; do
  for b in default tile; do
    if [ $b = tile ]; then export ED3DGS_LIB_PATH=$V; else unset ED3DGS_LIB_PATH; fi
    python bench.py --no-cpu-baseline --no-other-modes --steps 40 --warmup 10 > $out/bench_${b}_$rep.json 2> $out/bench_${b}_$rep.err
    python - <<PY
import json
d = json.load(open("$out/bench_${b}_$rep.json"))
k = {x["kernel"]: x for x in d["kernels"] if isinstance(x, dict) and "kernel" in x} if isinstance(d.get("kernels"), list) else d.get("kernels")
print("$b rep $rep ms/step", round(d["ms_per_step"], 4), "median", d["step_ms"]["median"], "K7", d["roofline_tile_backward"]["avg_launch_ms"], "records", d["roofline_tile_backward"]["valu_roof"].get("records_added_per_launch"))
PY
  done
done
for b in default tile; do
  if [ $b = tile ]; then export ED3DGS_LIB_PATH=$V; else unset ED3DGS_LIB_PATH; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_${b}_$c -o r --output-format csv -- python bench.py --no-cpu-baseline --no-other-modes --train-only --steps 4 --warmup 1 > $out/pmc_${b}_$c.log 2>&1
  done
  python tools/pmc_summary.py $out/pmc_${b}_FETCH_SIZE $out/pmc_${b}_WRITE_SIZE $out/pmc_${b}_fetch_write.md $out/pmc_${b}_summary.json > /dev/null 2>&1
  grep -E "render_backward|render_forward" $out/pmc_${b}_fetch_write.md
done
