#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lbs_prof -- python tools/bench_lbs.py > gpurun_out/lbs_prof.log 2>&1 || exit 1
f=$(ls gpurun_out/lbs_prof/*/*kernel_stats.csv | head -1)
cut -c1-160 $f | head -8
