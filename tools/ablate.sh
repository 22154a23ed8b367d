#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in 0 1 2 4 8 15; do
  export ED3DGS_FB_ABLATE=$a
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/abl_$a -o r --output-format csv -- python bench.py --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/abl_$a.log 2>&1
  echo "ablate=$a: $(python tools/summarize_prof.py gpurun_out/abl_$a | grep bwd_head | cut -d'|' -f4-5)"
done
