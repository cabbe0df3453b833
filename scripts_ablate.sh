#!/bin/bash
# timing ablations of the blend kernel + a PMC pass (profiling aid; outputs of the ablated runs are wrong by design)
mkdir -p gpurun_out
for fl in 0 1 2 3 4 7; do
  AMAV_RASTER_DEBUG=$fl timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('flags=$fl', 'ms_per_step=%.3f' % d['ms_per_step'], 'render_ms=%.3f' % d['roofline']['avg_launch_ms'])"
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc1 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc1.log 2>&1
echo pmc rc=$?
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc1/*/*counter_collection.csv')
print(f)
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f[0])):
        k = row['Kernel_Name'][:40]
        agg[k][row['Counter_Name']] += float(row['Counter_Value'])
    for k, v in agg.items():
        if 'render' in k or 'bin' in k or 'skin' in k or 'project' in k:
            print(k, {a: '%.3g' % b for a, b in v.items()})
PY
