#!/usr/bin/env python3
"""Does the split-product transformer step change what the autoregressive chain produces?
Rolls the full-size audio net of bench.py's full path (8 layers, S = 6304, random weights, proj_out scaled so the chain
stays bounded) over a whole clip in three processes and compares the tokens of every window:
    fp16 x 2 split products (default)            vs   exact-product fp32 kernels (AMAV_GEMM=f32 AMAV_ATTN=f32)
    exact-product fp32, library-default GEMMs    vs   exact-product fp32, tuned GEMMs        <- the yardstick: two
        fp32 runs that differ only in which library GEMM kernel (summation order) they use
usage: python tools/rollout_drift.py [frames, default 250]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def roll(frames, path):
    import numpy as np
    import torch

    sys.argv = ["bench.py", "--workload", "full", "--frames", str(frames), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    import bench

    args = bench.parse()
    with torch.no_grad():
        fp = bench.FullPath(args, "cuda:0", 0, frames)
        tri, smpl = fp.tokens()
        torch.cuda.synchronize()
    np.savez(path, tri=tri.float().cpu().numpy(), smpl=smpl.float().cpu().numpy())


def main():
    import numpy as np

    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 250
    import tempfile

    out = tempfile.mkdtemp(prefix="amav_drift_")  # ~200 MB of tokens: not under gpurun_out/ (64 MiB are copied back)
    variants = {"split": {}, "f32": {"AMAV_GEMM": "f32", "AMAV_ATTN": "f32"},
                "f32_default_gemms": {"AMAV_GEMM": "f32", "AMAV_ATTN": "f32", "AMAV_TUNED_GEMMS": "0"}}
    for name, env in variants.items():
        e = dict(os.environ, **env)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--roll", str(frames), os.path.join(out, f"drift_{name}.npz")],
                       check=True, env=e)
    data = {n: np.load(os.path.join(out, f"drift_{n}.npz")) for n in variants}
    scale = float(np.abs(data["f32"]["tri"]).max())
    print(f"{frames} frames; |triplane tokens| max {scale:.3f}")
    for a, b in (("split", "f32"), ("f32_default_gemms", "f32")):
        for key in ("tri", "smpl"):
            d = np.abs(data[a][key].astype(np.float64) - data[b][key])  # [1, frames, ...]
            per_frame = d.reshape(d.shape[1], -1).max(axis=1)
            marks = [0, min(5, frames - 1), min(49, frames - 1), frames - 1]
            print(f"{a:18s} vs {b}: {key:4s} max abs {d.max():.3e} (mean {d.mean():.3e}); by frame "
                  + ", ".join(f"#{m + 1}: {per_frame[m]:.2e}" for m in marks))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--roll":
        roll(int(sys.argv[2]), sys.argv[3])
    else:
        main()
