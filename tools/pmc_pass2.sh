#!/bin/bash
# usage: tools/pmc_pass2.sh <tag> <kernel-name filter regex> <counters...>  -- one rocprofv3 --pmc pass (kernel-trace only) over a
# short default-mode bench run; prints the per-dispatch means of the kernels whose name matches the filter
tag=$1; shift; filt=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_$tag -o r --output-format csv -- python bench.py --no-cpu-baseline --no-other-modes --steps 4 --warmup 1 > gpurun_out/pmc_$tag.log 2>&1
FILT="$filt" python - <<PY
import csv, glob, collections, os, re
f = glob.glob("gpurun_out/pmc_$tag/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter csv"); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if re.search(os.environ["FILT"], k):
        print(k, {c: "%d x %.3g" % (len(v), sum(v) / len(v)) for c, v in d.items()})
PY
