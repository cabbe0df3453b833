#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/ on the GPU box; tools/collect_profiles.py then writes the
# summaries judged under profiles/.  One program per rocprofv3 call, counters in their own passes.
#   usage (from the repo root, through gpurun): tools/gpu_profiles.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/final_*
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final_stats.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/final_pmc_$c -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/final_pmc_$c.log 2>&1 || exit 1
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_attn -- python tools/bench_attention.py > gpurun_out/final_attn.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err || exit 1
tail -c 400 gpurun_out/final_bench.json
