#!/bin/bash
# usage: tools/prof_bench.sh <tag> [bench args]   -- rocprofv3 kernel stats of bench.py, summary under gpurun_out/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o r --output-format csv -- python bench.py --no-cpu-baseline "$@" > gpurun_out/prof_$tag.log 2>&1
python tools/summarize_prof.py gpurun_out/prof_$tag gpurun_out/prof_${tag}_summary.md | head -${TOPN:-14}
grep -E '^\{' gpurun_out/prof_$tag.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'it/s', d['value'], 'fps', d['render_fps'])"
