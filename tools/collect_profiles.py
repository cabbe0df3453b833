#!/usr/bin/env python3
"""Copies the summaries of tools/gpu_profiles.sh (gpurun_out/final_*) into profiles/ under this round's names.
usage: python tools/collect_profiles.py [round-tag, default r02]"""
import collections
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
out = os.path.join(ROOT, "profiles")


def one(pattern, required=True):
    hits = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", pattern)), key=os.path.getmtime)
    if not hits:
        if required:
            raise SystemExit(f"missing {pattern}: run tools/gpu_profiles.sh through gpurun first")
        return None
    return hits[-1]  # gpurun merges every call's files into gpurun_out/: take the newest


def short(name):
    return name.split("(")[0].replace("void ", "")[:70]


def counters(pattern):
    """{kernel: {counter: mean per launch}, ...}, {kernel: launches}"""
    path = one(pattern, required=False)
    if path is None:
        return {}, {}
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        rows[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return ({k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in rows.items()},
            {k: max(len(v) for v in cs.values()) for k, cs in rows.items()})


shutil.copy(one("final_stats/*/*kernel_stats.csv"), os.path.join(out, f"{tag}_bench_render_kernel_stats.csv"))
shutil.copy(one("final_attn/*/*kernel_stats.csv"), os.path.join(out, f"{tag}_attention_transformer_kernel_stats.csv"))
ref = one("final_refiner/*/*kernel_stats.csv", required=False)
if ref:
    shutil.copy(ref, os.path.join(out, f"{tag}_point_refiner_kernel_stats.csv"))
src = one("final_bench.json", required=False)
if src:
    shutil.copy(src, os.path.join(out, f"{tag}_bench_render.json"))

# HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes)
fetch, n1 = counters("final_pmc_FETCH_SIZE/*/*counter_collection.csv")
write, n2 = counters("final_pmc_WRITE_SIZE/*/*counter_collection.csv")
with open(os.path.join(out, f"{tag}_bench_render_pmc_hbm.csv"), "w") as fh:
    fh.write("kernel,launches,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch\n")
    ks = sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, {}).get("FETCH_SIZE", 0) + write.get(k, {}).get("WRITE_SIZE", 0)))
    for k in ks:
        fh.write(f"{k},{max(n1.get(k, 0), n2.get(k, 0))},{fetch.get(k, {}).get('FETCH_SIZE', 0):.1f},"
                 f"{write.get(k, {}).get('WRITE_SIZE', 0):.1f}\n")


def merged(patterns, path, keep=None):
    table, launches = collections.defaultdict(dict), {}
    for pat in patterns:
        t, n = counters(pat)
        for k, cs in t.items():
            table[k].update(cs)
            launches[k] = n[k]
    names = sorted({c for cs in table.values() for c in cs})
    with open(path, "w") as fh:
        fh.write("kernel,launches," + ",".join(names) + "\n")
        for k in sorted(table, key=lambda k: -table[k].get("SQ_WAVE_CYCLES", table[k].get("SQ_BUSY_CYCLES", 0))):
            if keep and not any(s in k for s in keep):
                continue
            fh.write(f"{k},{launches[k]}," + ",".join(f"{table[k].get(c, float('nan')):.0f}" for c in names) + "\n")


# SQ counters of the render workload's kernels (three passes) and of the transformer step (two passes), per launch
merged(["final_pmc_sqA/*/*counter_collection.csv", "final_pmc_sqB/*/*counter_collection.csv",
        "final_pmc_sqC/*/*counter_collection.csv"], os.path.join(out, f"{tag}_bench_render_pmc_sq.csv"),
       keep=("amav::", "_ZN4amav"))
merged(["final_attn_pmc/*/*counter_collection.csv", "final_attn_pmc2/*/*counter_collection.csv"],
       os.path.join(out, f"{tag}_attention_transformer_pmc_sq.csv"), keep=("amav::", "_ZN4amav", "Cijk", "gemm", "Gemm"))
merged(["final_refiner_pmc/*/*counter_collection.csv"], os.path.join(out, f"{tag}_point_refiner_pmc_sq.csv"),
       keep=("amav::", "_ZN4amav", "Cijk"))
print("wrote", sorted(os.listdir(out)))
