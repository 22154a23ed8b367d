#!/bin/bash
# Round 4, session 2, call A: K7 bank-masked reduce (parity + A/B against -DED3_K7_BANK_REDUCE=0), head-split probe, CU-mask probe
out=gpurun_out/r4s2a; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== parity (default build: bank-masked reduce)"
timeout -k 10 600 python -m pytest tests/test_raster_parity_gpu.py tests/test_fullsize_gpu.py tests/test_odd_sizes_gpu.py tests/test_reference_paths_gpu.py tests/test_render_variants_gpu.py -q -m gpu -x > $out/pytest.log 2>&1; tail -3 $out/pytest.log
V=$GRAFT_REPO_ROOT/e-d3dgs_amd/csrc/variants/libed3dgs_hip_k7oldreduce.so
for rep in 1 2 3; do
  for b in default old; do
    if [ $b = old ]; then export ED3DGS_LIB_PATH=$V; else unset ED3DGS_LIB_PATH; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-modes --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
print('$b rep $rep: ms/step %.4f median %.4f | K7 %.4f ms' % (d['ms_per_step'], d['step_ms']['median'], d['roofline_tile_backward']['avg_launch_ms']))"
  done
done
unset ED3DGS_LIB_PATH
echo "== head split probe"
timeout -k 10 300 python tools/r04_head_split_probe.py 2>&1 | tee $out/head_split.txt
echo "== CU mask probe"
for cfg in "256 all" "32 contig" "32 strided" "224 contig" "224 strided"; do
  set -- $cfg
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/cu_$1_$2 -o r --output-format csv -- python tools/r04_cu_mask_probe.py $1 $2 > $out/cu_$1_$2.log 2>&1
  grep -E "forward render|rc " $out/cu_$1_$2.log
  python - <<PY
import csv
rows = list(csv.DictReader(open("$out/cu_$1_$2/r_kernel_stats.csv")))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("deform_forward", "preprocess_kernel", "rocprim", "bin2_", "bin_tiles", "render_forward")):
        print("   %-70s calls %4s avg %9.1f us" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
