#!/usr/bin/env python3
"""GPU busy fraction of a rocprofv3 --kernel-trace csv directory: union of kernel intervals / (last end - first start),
optionally restricted to the last `frac` of the trace (steady state).   python tools/gpu_busy.py <dir> [frac]"""
import csv
import glob
import sys

d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
iv = []
for path in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
iv.sort()
t0, t1 = iv[0][0], max(e for _, e, _ in iv)
cut = t1 - (t1 - t0) * frac
iv = [x for x in iv if x[0] >= cut]
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e, _ in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = cur_e - iv[0][0]
gaps.sort()
print(f"kernels {len(iv)}, span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms = {busy / span:.3f}; "
      f"gaps: n {len(gaps)}, total {sum(gaps) / 1e6:.2f} ms, median {gaps[len(gaps) // 2] / 1e3:.1f} us, "
      f"p99 {gaps[int(len(gaps) * 0.99)] / 1e3:.1f} us, max {gaps[-1] / 1e3:.1f} us")
