"""Probe: an fp32-equivalent GEMM as ONE bf16 library GEMM over split operands.
x = x1 + x2 + x3 (bf16 each), W likewise; x W^T ~= sum over the six partial products with i + j <= 4, laid out along K:
A' = [x3 x2 x1 x2 x1 x1], B' = [W1 W2 W3 W1 W2 W1] (small terms first).  Prints time and error against fp64."""
import sys
import time

import torch


def split3(x):
    a = x.to(torch.bfloat16)
    r = x - a.float()
    b = r.to(torch.bfloat16)
    c = (r - b.float()).to(torch.bfloat16)
    return a, b, c


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
    for (M, K, N) in [(6304, 512, 4096), (6304, 2048, 512), (6304, 512, 1536), (6304, 512, 512)]:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * K ** -0.5
        bias = torch.randn(N, device=dev)
        ref = (x.double() @ w.double().t() + bias.double())
        y32 = torch.addmm(bias, x, w.t())
        x1, x2, x3 = split3(x)
        w1, w2, w3 = split3(w)
        A = torch.cat([x3, x2, x1, x2, x1, x1], dim=1).contiguous()
        B = torch.cat([w1, w2, w3, w1, w2, w1], dim=1).contiguous()  # [N, 6K]
        try:
            y = torch.mm(A, B.t(), out_dtype=torch.float32) + bias
            mode = "out_dtype"
            f = lambda: torch.mm(A, B.t(), out_dtype=torch.float32)
        except Exception as e:  # noqa: BLE001
            print("out_dtype unsupported:", str(e)[:100])
            return
        t32 = bench(lambda: torch.addmm(bias, x, w.t()))
        tsp = bench(f)
        tcat = bench(lambda: torch.cat([x3, x2, x1, x2, x1, x1], dim=1))
        flop = 2.0 * M * K * N
        print("M=%d K=%d N=%d: fp32 %.3f ms (%.0f TF/s) err %.2e | split %.3f ms (%.0f TF/s fp32-equiv) err %.2e | cat %.3f ms [%s]"
              % (M, K, N, t32, flop / t32 / 1e9, (y32 - ref).abs().max().item(), tsp, flop / tsp / 1e9,
                 (y - ref).abs().max().item(), tcat, mode))


if __name__ == "__main__":
    main()
