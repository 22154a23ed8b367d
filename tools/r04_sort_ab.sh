#!/bin/bash
# Round 4: binning level 1, hand-written bucket + rank sort (ED3DGS_SORT_HANDWRITTEN=1) vs the library's sort (default), one box; the trace is of the hand-written one.
out=gpurun_out/r4e; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_binning_stress_gpu.py tests/test_raster_parity_gpu.py tests/test_fullsize_gpu.py tests/test_integrate_gpu.py -q -m gpu -x > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for rep in 1 2; do
  for v in rank library; do
    if [ $v = rank ]; then export ED3DGS_SORT_HANDWRITTEN=1; else unset ED3DGS_SORT_HANDWRITTEN; fi
    python bench.py --no-cpu-baseline --no-other-modes --steps 40 --warmup 10 > $out/bench_${v}_$rep.json 2> $out/bench_${v}_$rep.err
    python -c "
import json; d=json.load(open('$out/bench_${v}_$rep.json')); print('$v rep $rep ms/step %.4f median %.4f fps %.1f' % (d['ms_per_step'], d['step_ms']['median'], d['render_fps']))"
  done
done
export ED3DGS_SORT_HANDWRITTEN=1
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/trace -o r --output-format csv -- python bench.py --no-cpu-baseline --no-other-modes --train-only --steps 6 --warmup 2 > $out/trace.log 2>&1
python tools/step_trace.py $out/trace > $out/step_trace.txt 2>&1; grep -E "depth_rank|preprocess_kernel|bin2_countA|launches" $out/step_trace.txt | cut -c1-150
