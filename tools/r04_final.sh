#!/bin/bash
# round-4 closing run (one gpurun call): the whole GPU suite, smoke(), the profile set (tools/r04_profiles.sh), a 2-rank rehearsal on the one GPU
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r4final; mkdir -p $o
python -m pytest tests -m gpu -q > $o/pytest.log 2>&1; tail -3 $o/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1; tail -1 $o/smoke.log
tools/r04_profiles.sh r04 > $o/profiles.log 2>&1; grep -E "ms/step|launches|timed region" $o/profiles.log gpurun_out/r04_bench.err | tail -5
ED3DGS_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline > $o/bench_2rank_gloo.json 2> $o/bench_2rank_gloo.err; echo "2-rank rc=$?"; tail -c 600 $o/bench_2rank_gloo.json
