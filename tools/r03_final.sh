#!/bin/bash
# round-3 closing run (one gpurun call): the whole GPU suite, smoke(), the default bench line, a 2-rank rehearsal on the one GPU
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3final; mkdir -p $o
python -m pytest tests -m gpu -q > $o/pytest.log 2>&1; tail -3 $o/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1; tail -1 $o/smoke.log
python bench.py > $o/bench.json 2> $o/bench.err; tail -2 $o/bench.err
ED3DGS_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline > $o/bench_2rank_gloo.json 2> $o/bench_2rank_gloo.err; echo "2-rank rc=$?"; tail -c 1500 $o/bench_2rank_gloo.json
