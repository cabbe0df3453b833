#!/bin/bash
# usage: scripts_gpu_round.sh <tag>   -- tests, smoke, bench; stops at the first step that hangs (timeout)
tag=$1
mkdir -p gpurun_out
step() { # name, seconds, command...
  name=$1; secs=$2; shift 2
  timeout -k 10 $secs "$@" > gpurun_out/${name}_${tag}.log 2>&1
  rc=$?
  echo "[$name] rc=$rc"
  tail -n 15 gpurun_out/${name}_${tag}.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] timed out: stopping"; exit 1; fi
  return 0
}
step pytest 500 python -m pytest tests -m gpu -q || exit 1
step smoke 200 python -c "import __graft_entry__ as g; g.smoke()" || exit 1
step bench 400 python bench.py --steps 10 --warmup 2 || exit 1
if [ "$2" = "prof" ]; then
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_${tag}.log 2>&1
  echo "[prof] rc=$?"
  f=$(ls gpurun_out/prof_${tag}/*/*kernel_stats.csv | head -1)
  cut -c1-150 $f | head -14
fi
