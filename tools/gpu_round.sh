#!/bin/bash
# usage: tools/gpu_round.sh <tag> [prof]  -- GPU tests, smoke, bench; stops at the first step that hangs (timeout)
tag=$1
mkdir -p gpurun_out
rm -f gpurun_out/raster_parity.jsonl
step() { # name, seconds, command...
  name=$1; secs=$2; shift 2
  timeout -k 10 $secs "$@" > gpurun_out/${name}_${tag}.log 2>&1
  rc=$?
  echo "[$name] rc=$rc"
  tail -n 12 gpurun_out/${name}_${tag}.log | cut -c1-400
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] timed out: stopping"; exit 1; fi
  if grep -q "Memory access fault" gpurun_out/${name}_${tag}.log; then echo "[$name] GPU fault: stopping"; exit 1; fi
  return 0
}
step pytest 800 python -m pytest tests -m gpu -q -x --durations=8 || exit 1
step smoke 200 python -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 500 python bench.py > gpurun_out/bench_${tag}.json 2> gpurun_out/bench_${tag}.err
echo "[bench] rc=$?"; tail -c 300 gpurun_out/bench_${tag}.err
python - <<PY
import json
d = json.load(open("gpurun_out/bench_${tag}.json"))
fp = d.get("full_path", {})
print("render", round(d["value"]), "frames/s", d["ms_per_step"], d["step_device_ms"], "blend frac", d["roofline"]["frac"])
print("parity", json.dumps(d.get("parity"))[:600])
print("full_path", fp.get("value"), "frames/s", fp.get("ms_per_step"), json.dumps(fp.get("roofline"))[:700])
print("full parity", json.dumps(fp.get("parity"))[:700], json.dumps(fp.get("cpu_baseline"))[:400])
PY
