#!/bin/bash
# Round-4 profile artefacts, one gpurun call.  Every profiled run is `bench.py --train-only --no-other-modes`: every launch in it is
# a training-step launch (round 3's PMC passes mixed in the non-keeping inference launches of the render-fps loop).
# usage (GPU box, repo root): tools/r04_profiles.sh [tag]     -> gpurun_out/<tag>_*; copy what is to be judged into profiles/
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CMD="python bench.py --no-cpu-baseline --no-other-modes --train-only --steps 4 --warmup 1"
echo "== bench line (full default run: the numbers the summaries are read beside)"; python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -2 gpurun_out/${tag}_bench.err
echo "== kernel stats"; TOPN=40 tools/prof_bench.sh ${tag}_stats --no-other-modes --train-only --steps 40 --warmup 5 > gpurun_out/${tag}_stats.txt 2>&1; tail -3 gpurun_out/${tag}_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== $c"; timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_${tag}_$c -o r --output-format csv -- $CMD > gpurun_out/pmc_${tag}_$c.log 2>&1
done
python tools/pmc_summary.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE gpurun_out/${tag}_pmc_fetch_write.md gpurun_out/${tag}_pmc_summary.json "$CMD" gpurun_out/${tag}_bench.json > /dev/null 2>&1
grep -E "deform_|render_|preprocess" gpurun_out/${tag}_pmc_fetch_write.md | cut -c1-200
echo "== SQ"; timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d gpurun_out/pmc_${tag}_sq -o r --output-format csv -- $CMD > gpurun_out/pmc_${tag}_sq.log 2>&1
python tools/sq_table.py gpurun_out/pmc_${tag}_sq gpurun_out/prof_${tag}_stats gpurun_out/${tag}_pmc_sq.md ${tag} > /dev/null 2>&1; head -30 gpurun_out/${tag}_pmc_sq.md | cut -c1-220
echo "== trace"; timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/${tag}_trace -o r --output-format csv -- python bench.py --no-cpu-baseline --no-other-modes --train-only --steps 6 --warmup 2 > gpurun_out/${tag}_trace.log 2>&1; python tools/step_trace.py gpurun_out/${tag}_trace > gpurun_out/${tag}_step_trace.txt 2>&1; tail -1 gpurun_out/${tag}_step_trace.txt
