#!/bin/bash
# Round-3 profile artefacts, one gpurun call: kernel stats, FETCH / WRITE passes, SQ pass, step trace, full bench line.
# usage (on the GPU box, from the repo root): tools/r03_profiles.sh <tag>
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== kernel stats (every launch of the run is a training-step launch: --train-only)"; TOPN=40 tools/prof_bench.sh ${tag}_stats --no-other-modes --train-only --steps 40 --warmup 5 > gpurun_out/${tag}_stats.txt 2>&1; tail -3 gpurun_out/${tag}_stats.txt
echo "== FETCH_SIZE"; tools/pmc_pass.sh ${tag}_fetch FETCH_SIZE > gpurun_out/${tag}_fetch.txt 2>&1
echo "== WRITE_SIZE"; tools/pmc_pass.sh ${tag}_write WRITE_SIZE > gpurun_out/${tag}_write.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write gpurun_out/${tag}_pmc_fetch_write.md gpurun_out/${tag}_pmc_summary.json > /dev/null 2>&1
echo "== SQ"; tools/pmc_pass2.sh ${tag}_sq "deform_|render_|preprocess" SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES > gpurun_out/${tag}_sq.txt 2>&1
echo "== trace"; timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/${tag}_trace -o r --output-format csv -- python bench.py --no-cpu-baseline --no-other-modes --steps 6 --warmup 2 > gpurun_out/${tag}_trace.log 2>&1; python tools/step_trace.py gpurun_out/${tag}_trace > gpurun_out/${tag}_step_trace.txt 2>&1; tail -1 gpurun_out/${tag}_step_trace.txt
echo "== bench"; python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -2 gpurun_out/${tag}_bench.err; head -c 600 gpurun_out/${tag}_bench.json
