#!/bin/bash
# Round 4 A/B on one box: activations inside the deformation kernels vs a launch of their own; step trace of the default.
out=gpurun_out/r4d; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in fused separate; do
    if [ $v = separate ]; then export ED3DGS_SEPARATE_ACTIVATIONS=1; else unset ED3DGS_SEPARATE_ACTIVATIONS; fi
    python bench.py --no-cpu-baseline --no-other-modes --steps 40 --warmup 10 > $out/bench_${v}_$rep.json 2> $out/bench_${v}_$rep.err
    python -c "
import json; d=json.load(open('$out/bench_${v}_$rep.json')); print('$v rep $rep ms/step %.4f median %.4f fps %.1f' % (d['ms_per_step'], d['step_ms']['median'], d['render_fps']))"
  done
done
unset ED3DGS_SEPARATE_ACTIVATIONS
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/trace -o r --output-format csv -- python bench.py --no-cpu-baseline --no-other-modes --train-only --steps 6 --warmup 2 > $out/trace.log 2>&1
python tools/step_trace.py $out/trace > $out/step_trace.txt 2>&1; tail -45 $out/step_trace.txt
