#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_sq -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_sq.log 2>&1
echo rc=$?
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/pmc_sq2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_sq2.log 2>&1
echo rc=$?
python - <<'PY'
import csv, glob, collections
for d in ("pmc_sq", "pmc_sq2"):
    f = glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")
    if not f: print(d, "missing"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f[0])):
        agg[row["Kernel_Name"].split("(")[0][-40:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        if "amav" in k:
            print(k, {a: "%.4g" % (sum(b) / len(b)) for a, b in v.items()})
PY
