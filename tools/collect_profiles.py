#!/usr/bin/env python3
"""Copies the summaries of tools/gpu_profiles.sh (gpurun_out/final_*) into profiles/ under this round's names.
usage: python tools/collect_profiles.py [round-tag, default r01]"""
import csv
import glob
import collections
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = os.path.join(ROOT, "profiles")


def one(pattern):
    hits = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", pattern)), key=os.path.getmtime)
    if not hits:
        raise SystemExit(f"missing {pattern}: run tools/gpu_profiles.sh through gpurun first")
    return hits[-1]  # gpurun merges every call's files into gpurun_out/: take the newest


shutil.copy(one("final_stats/*/*kernel_stats.csv"), os.path.join(out, f"{tag}_bench_render_kernel_stats.csv"))
shutil.copy(one("final_attn/*/*kernel_stats.csv"), os.path.join(out, f"{tag}_attention_transformer_kernel_stats.csv"))
shutil.copy(one("final_bench.json"), os.path.join(out, f"{tag}_bench_render.json"))
per = collections.defaultdict(lambda: {"n": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(one(f"final_pmc_{counter}/*/*counter_collection.csv"))):
        if r["Counter_Name"] == counter:
            rows[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
    for k, v in rows.items():
        per[k]["n"] = len(v)
        per[k][counter] = sum(v) / len(v)
with open(os.path.join(out, f"{tag}_bench_render_pmc_hbm.csv"), "w") as fh:
    fh.write("kernel,launches,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch\n")
    for k, v in sorted(per.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"])):
        fh.write(f"{k},{v['n']},{v['FETCH_SIZE']:.1f},{v['WRITE_SIZE']:.1f}\n")
print("wrote", sorted(os.listdir(out)))
