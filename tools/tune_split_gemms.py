#!/usr/bin/env python3
"""Finds, for every fp16 x 2 split projection shape of the transformer step, the fastest hipBLASLt kernel of the library
build that is loaded in this process (torch's own copy), and writes audio-motion-avatar_amd/gemm_split_tuning_gfx950.csv
(rows, n, k3, algorithm index, ms, heuristic ms; first line: the library version the indices belong to).

    python tools/tune_split_gemms.py [rows ...]        default rows: 6304 (S of the audio net)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_motion_avatar_amd import ops  # noqa: E402

rows_list = [int(x) for x in sys.argv[1:]] or [6304]
shapes = [(1536, 1536), (512, 1536), (4096, 1536), (512, 6144)]  # (n, k3): q/k/v, to_out, GEGLU projection, ff out
out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio-motion-avatar_amd", "gemm_split_tuning_gfx950.csv")
lines = [f"# {ops.gemm_library_version()} {torch.cuda.get_device_properties(0).gcnArchName.split(':')[0]}"]
g = torch.Generator(device="cuda").manual_seed(0)
for rows in rows_list:
    for n, k3 in shapes:
        COPIES = 8  # one operand set per transformer layer: a kernel is timed as it runs in the step, not out of a hot L2
        a8 = (torch.randn(COPIES, rows, k3, device="cuda", generator=g) * 100).half()   # real magnitudes (zeros run at other clocks)
        w8 = (torch.randn(COPIES, n, k3, device="cuda", generator=g) * 100).half()
        a, w = a8[0], w8[0]
        t0 = time.time()
        idx, best, heur = ops.gemm_split_fp16_tune(a8, w8, repeats=16)
        # confirm against torch's own path on the same operands
        ref = torch.mm(a, w.t(), out_dtype=torch.float32)
        got = ops.gemm_split_fp16(a, w, 1.0, idx)
        err = float((got - ref).abs().max() / ref.abs().max())
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for _ in range(3):
            torch.mm(a, w.t(), out_dtype=torch.float32)
        ev[0].record()
        for _ in range(20):
            torch.mm(a, w.t(), out_dtype=torch.float32)
        ev[1].record()
        torch.cuda.synchronize()
        torch_ms = ev[0].elapsed_time(ev[1]) / 20
        print(f"rows {rows} n {n} k3 {k3}: best index {idx} {best:.4f} ms, heuristic {heur:.4f} ms, torch.mm {torch_ms:.4f} ms, "
              f"rel diff vs torch {err:.1e} ({time.time() - t0:.0f} s)", flush=True)
        lines.append(f"{rows},{n},{k3},{idx},{best:.5f},{heur:.5f}")
with open(out_path, "w") as fh:
    fh.write("\n".join(lines) + "\n")
print("wrote", out_path)
