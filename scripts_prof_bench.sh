#!/bin/bash
# rocprofv3 kernel stats of the default bench (short), printed
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$1 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_$1.log 2>&1 || exit 1
f=$(ls gpurun_out/prof_$1/*/*kernel_stats.csv | head -1)
cut -c1-60,90-200 $f | head -14
