#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/ on the GPU box; tools/collect_profiles.py then writes the
# summaries judged under profiles/.  One program per rocprofv3 call; counters in their own passes (never combined with
# trace domains other than --kernel-trace).
#   usage (from the repo root, through gpurun): tools/gpu_profiles.sh [render|attn|refiner|all]
what=${1:-all}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-full-path --no-refiner --no-extra-configs"
pass() { # dir, seconds, rocprof args..., -- program
  d=$1; secs=$2; shift 2
  rm -rf gpurun_out/$d
  timeout -k 10 $secs rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$d "$@" > gpurun_out/$d.log 2>&1
  rc=$?
  echo "[$d] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi
}
if [ "$what" = "render" ] || [ "$what" = "all" ]; then
  pass final_stats 300 --stats -- $BENCH
  pass final_pmc_FETCH_SIZE 300 --pmc FETCH_SIZE -- $BENCH
  pass final_pmc_WRITE_SIZE 300 --pmc WRITE_SIZE -- $BENCH
  pass final_pmc_sqA 300 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -- $BENCH
  pass final_pmc_sqB 300 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS_F32 -- $BENCH
  pass final_pmc_sqC 300 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_CYCLES SQ_INSTS_BRANCH -- $BENCH
fi
if [ "$what" = "attn" ] || [ "$what" = "all" ]; then
  pass final_attn 300 --stats -- python tools/bench_attention.py
  pass final_attn_pmc 300 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -- python tools/bench_attention.py
  pass final_attn_pmc2 300 --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES -- python tools/bench_attention.py
fi
if [ "$what" = "refiner" ] || [ "$what" = "all" ]; then
  pass final_refiner 300 --stats -- python tools/bench_refiner.py 8 8
  pass final_refiner_pmc 300 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -- python tools/bench_refiner.py 8 8
fi
ls gpurun_out/final_*/*/ 2>/dev/null | head -40
