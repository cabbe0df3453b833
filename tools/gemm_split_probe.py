"""Probe: fp32-equivalent GEMMs as ONE low-precision library GEMM over split operands.
  bf16 x 3: x = x1 + x2 + x3, six partial products (i + j <= 4), K' = 6 K
  fp16 x 2: x 2^e = h1 + h2 (power-of-two pre-scale keeps the residuals out of fp16's subnormals), three partial
            products (h1 g1, h1 g2, h2 g1), K' = 3 K, result rescaled by a power of two
Prints time and max abs error against fp64 next to the library's fp32 GEMM."""
import math
import time

import torch


def split_bf16(x):
    a = x.to(torch.bfloat16)
    r = x - a.float()
    b = r.to(torch.bfloat16)
    c = (r - b.float()).to(torch.bfloat16)
    return a, b, c


def split_fp16(x, bound):
    e = math.floor(math.log2(32768.0 / bound))
    xs = x * 2.0 ** e
    a = xs.to(torch.float16)
    b = (xs - a.float()).to(torch.float16)
    return a, b, e


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    dev = "cuda"
    torch.manual_seed(0)
    for (M, K, N) in [(6304, 512, 4096), (6304, 512, 1536), (6304, 512, 512), (6304, 2048, 512)]:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * K ** -0.5
        ref = x.double() @ w.double().t()
        y32 = x @ w.t()
        x1, x2, x3 = split_bf16(x)
        w1, w2, w3 = split_bf16(w)
        A = torch.cat([x3, x2, x1, x2, x1, x1], dim=1).contiguous()
        B = torch.cat([w1, w2, w3, w1, w2, w1], dim=1).contiguous()
        y3 = torch.mm(A, B.t(), out_dtype=torch.float32)
        h1, h2, ex = split_fp16(x, math.sqrt(K))           # LayerNorm bound: |x| <= sqrt(K - 1) before the affine
        g1, g2, ew = split_fp16(w, w.abs().max().item())
        A2 = torch.cat([h2, h1, h1], dim=1).contiguous()
        B2 = torch.cat([g1, g2, g1], dim=1).contiguous()
        y2 = torch.mm(A2, B2.t(), out_dtype=torch.float32) * 2.0 ** -(ex + ew)
        dummy = torch.zeros(N, device=dev)
        try:
            y2b = torch.addmm(dummy, A2, B2.t(), beta=0.0, alpha=2.0 ** -(ex + ew), out_dtype=torch.float32)
            alpha_ok = (y2b - y2).abs().max().item()
            t2a = bench(lambda: torch.addmm(dummy, A2, B2.t(), beta=0.0, alpha=2.0 ** -(ex + ew), out_dtype=torch.float32))
        except Exception as e:  # noqa: BLE001
            alpha_ok, t2a = str(e)[:80], float("nan")
        t32 = bench(lambda: x @ w.t())
        t3 = bench(lambda: torch.mm(A, B.t(), out_dtype=torch.float32))
        t2 = bench(lambda: torch.mm(A2, B2.t(), out_dtype=torch.float32))
        err = lambda y: (y.double() - ref).abs().max().item()
        print("M=%d K=%d N=%d: fp32 %.3f ms err %.2e | bf16x3 %.3f ms err %.2e | fp16x2 %.3f ms err %.2e (scales 2^%d 2^%d)"
              " | addmm alpha: %.3f ms diff %s" % (M, K, N, t32, err(y32), t3, err(y3), t2, err(y2), ex, ew, t2a, alpha_ok))


if __name__ == "__main__":
    main()
