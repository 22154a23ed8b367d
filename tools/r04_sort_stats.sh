#!/bin/bash
# per-kernel average durations of a training run with either level-1 sort (which kernel pays for the shorter sort?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
for v in 0 1; do
  export ED3DGS_SORT_HANDWRITTEN=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r4h/prof_$v -o r --output-format csv -- python bench.py --no-cpu-baseline --no-other-modes --train-only --steps 60 --warmup 5 > gpurun_out/r4h/prof_$v.log 2>&1
done
python - <<'PY'
import csv, glob
t = {}
for v in (0, 1):
    f = glob.glob("gpurun_out/r4h/prof_%d/**/*kernel_stats.csv" % v, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        t.setdefault(r["Name"][:70], {})[v] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
tot = [0.0, 0.0]
for k, d in sorted(t.items(), key=lambda kv: -max(x[1] * x[0] for x in kv[1].values())):
    a, b = d.get(0, (0, 0.0)), d.get(1, (0, 0.0))
    per = 65.0
    ca, cb = a[0] * a[1] / per, b[0] * b[1] / per
    tot[0] += ca; tot[1] += cb
    if max(ca, cb) > 3.0:
        print("%-72s lib %3d x %7.1f us   rank %3d x %7.1f us   per step %7.1f -> %7.1f" % (k, a[0], a[1], b[0], b[1], ca, cb))
print("GPU busy per step (us): library %.1f, rank %.1f" % (tot[0], tot[1]))
PY
