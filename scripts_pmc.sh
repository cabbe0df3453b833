#!/bin/bash
# HBM traffic counters of the bench (separate --pmc passes, kernel-trace only), summarised per kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
  echo "$c rc=$?"
done
python - <<'PY'
import csv, glob, collections
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{c}/*/*counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] == c:
            agg[row["Kernel_Name"].split("(")[0][-60:]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        out[k][c] = (sum(v) / len(v), len(v))
with open("gpurun_out/pmc_summary.csv", "w") as fh:
    fh.write("kernel,launches,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch\n")
    for k, v in sorted(out.items(), key=lambda kv: -sum(x[0] for x in kv[1].values())):
        fs, ws = v.get("FETCH_SIZE", (0, 0)), v.get("WRITE_SIZE", (0, 0))
        line = f"{k},{max(fs[1], ws[1])},{fs[0]:.1f},{ws[0]:.1f}"
        fh.write(line + "\n")
        if "amav" in k:
            print(line)
PY
