#!/usr/bin/env python3
"""Diagnostic: time of the MFMA self-attention kernel and of one transformer step at the reference's shape."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_motion_avatar_amd import ops  # noqa: E402
from audio_motion_avatar_amd.transformer import Transformer1D_nn  # noqa: E402

B, S, H = 1, 6304, 8
qkv = torch.randn(B, S, 3 * H * 64, device="cuda")
i = H * 64
q, k, v = qkv[..., :i], qkv[..., i:2 * i], qkv[..., 2 * i:]
for _ in range(3):
    ops.selfattn(q, k, v, H)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    ops.selfattn(q, k, v, H)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
flop = 4.0 * S * S * 64 * H * B
print(f"selfattn S={S} H={H}: {dt * 1e3:.3f} ms  {flop / dt / 1e12:.1f} TFLOP/s fp32-equivalent (operand split + kernel + combine; fp32 MFMA peak 157)")

qc, kc, vc = (t.contiguous().view(B, S, H, 64).transpose(1, 2) for t in (q, k, v))
for _ in range(2):
    torch.nn.functional.scaled_dot_product_attention(qc, kc, vc)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    torch.nn.functional.scaled_dot_product_attention(qc, kc, vc)
torch.cuda.synchronize()
print(f"torch SDPA fp32 (library): {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms")

net = Transformer1D_nn(num_layers=8, attention_head_dim=64, in_channels=256, num_attention_heads=8,
                       cross_attention_dim=768).cuda().eval()
x = torch.randn(1, 256, S, device="cuda")
a = torch.randn(1, 1, 768, device="cuda")
with torch.no_grad():
    for _ in range(2):
        net(x, a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        net(x, a)
    torch.cuda.synchronize()
print(f"Transformer1D_nn step (8 layers, S={S}): {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
