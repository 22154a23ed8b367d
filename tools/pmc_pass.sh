#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counters...>  -- one rocprofv3 --pmc pass (kernel-trace only) over a short bench run
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_$tag -o r --output-format csv -- python bench.py --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/pmc_$tag.log 2>&1
ls gpurun_out/pmc_$tag | head
python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_$tag/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter csv"); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "render_" in k or "deform_" in k or "preprocess" in k:
        print(k, {c: (len(v), sum(v) / len(v)) for c, v in d.items()})
PY
