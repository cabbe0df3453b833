#!/usr/bin/env python3
"""Diagnostic: library fp32 GEMM time for the transformer's four projection shapes, two operand layouts."""
import time

import torch
import torch.nn.functional as F

S = 6304
shapes = [("qkv", 512, 1536), ("out", 512, 512), ("ff1", 512, 4096), ("ff2", 2048, 512)]


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for name, k, n in shapes:
    x = torch.randn(1, S, k, device="cuda")
    w = torch.randn(n, k, device="cuda")
    b = torch.randn(n, device="cuda")
    wt = w.t().contiguous()
    a = t(lambda: F.linear(x, w, b))
    c = t(lambda: torch.addmm(b, x[0], wt))
    flop = 2.0 * S * k * n
    print(f"{name}: F.linear {a * 1e6:7.1f} us ({flop / a / 1e12:5.1f} TF)   addmm(x, Wt) {c * 1e6:7.1f} us ({flop / c / 1e12:5.1f} TF)")

# the other BLAS backend, if this build lets us choose
for lib in ("cublas", "cublaslt"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:  # noqa: BLE001
        print(f"preferred_blas_library({lib!r}) unavailable: {e}")
        continue
    for name, k, n in shapes:
        x = torch.randn(1, S, k, device="cuda")
        w = torch.randn(n, k, device="cuda")
        b = torch.randn(n, device="cuda")
        a = t(lambda: F.linear(x, w, b))
        print(f"[{lib}] {name}: F.linear {a * 1e6:7.1f} us ({2.0 * S * k * n / a / 1e12:5.1f} TF)")
