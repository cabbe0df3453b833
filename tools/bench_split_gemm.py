#!/usr/bin/env python3
"""Diagnostic: the hand-written fp16 x 2 split GEMM (algo_index = -2) against the library GEMM over the K-concatenated
operands (heuristic and tuned index), for the transformer block's four projections at 6304 rows: agreement and time
(operands rotated over 8 sets, as inside the step)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_motion_avatar_amd import ops, tuning  # noqa: E402

M = 6304
g = torch.Generator(device="cuda").manual_seed(1)
for name, K, N in (("q/k/v", 512, 1536), ("to_out", 512, 512), ("geglu proj", 512, 4096), ("ff out", 2048, 512)):
    sets = 8
    A, W = [], []
    for _ in range(sets):
        h = torch.randn(M, K, device="cuda", generator=g) * 3.0
        w = torch.randn(N, K, device="cuda", generator=g) * 2.0
        h1, w1 = h.half(), w.half()
        h2, w2 = (h - h1.float()).half(), (w - w1.float()).half()
        A.append(torch.cat([h2, h1, h1], dim=1).contiguous())
        W.append(torch.cat([w1, w2, w1], dim=1).contiguous())
    ref = (A[0][:, K:2 * K].double() + A[0][:, :K].double()) @ (W[0][:, :K].double() + W[0][:, K:2 * K].double()).t()
    lib = ops.gemm_split_fp16(A[0], W[0], 1.0, -1)
    hw = ops.gemm_split_fp16(A[0], W[0], 1.0, -2)
    scale = ref.abs().max().item()
    print(f"{name:10s} K={K} N={N}: |hw - lib| {(hw - lib).abs().max().item() / scale:.2e}  |hw - fp64| {(hw.double() - ref).abs().max().item() / scale:.2e}"
          f"  |lib - fp64| {(lib.double() - ref).abs().max().item() / scale:.2e} (relative to max |out|)")
    idx = tuning.split_gemm_index(M, N, 3 * K)

    def timed(algo):
        for i in range(4):
            ops.gemm_split_fp16(A[i % sets], W[i % sets], 1.0, algo)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(40):
            ops.gemm_split_fp16(A[i % sets], W[i % sets], 1.0, algo)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 40

    flop = 2.0 * M * N * 3 * K
    for label, algo in (("library heuristic", -1), (f"library index {idx}", idx), ("hand-written", -2)):
        ms = timed(algo)
        print(f"    {label:24s} {ms * 1e3:7.1f} us  {flop / ms / 1e9:7.0f} TFLOP/s issued")
