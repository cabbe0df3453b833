#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --output-format csv directory: total / calls / mean per kernel name.

    python tools/kernel_stats.py gpurun_out/prof_x [top]
"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot, cnt = defaultdict(float), defaultdict(int)
for path in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"]
        tot[name] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
        cnt[name] += 1
all_us = sum(tot.values())
print(f"{'total us':>12s} {'calls':>7s} {'mean us':>10s} {'%':>6s}  kernel")
for name in sorted(tot, key=tot.get, reverse=True)[:top]:
    print(f"{tot[name]:12.1f} {cnt[name]:7d} {tot[name] / cnt[name]:10.2f} {100 * tot[name] / all_us:6.2f}  {name[:110]}")
print(f"{all_us:12.1f} us in {sum(cnt.values())} launches")
