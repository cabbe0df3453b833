#!/usr/bin/env python3
"""Diagnostic: error of the self-attention kernels (AMAV_ATTN=f32: fp32 MFMA; AMAV_ATTN=bf16: bf16 x 3 split; default: fp16 x 2 split) against fp64 SDPA
at the reference shape, next to the library's fp32 SDPA."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_motion_avatar_amd import ops  # noqa: E402

torch.manual_seed(0)
B, S, H = 1, 6304, 8
for scale_in in (1.0, 4.0):
    qkv = torch.randn(B, S, 3 * H * 64, device="cuda") * scale_in
    i = H * 64
    q, k, v = qkv[..., :i], qkv[..., i:2 * i], qkv[..., 2 * i:]
    out = ops.selfattn(q, k, v, H)
    heads = lambda t: t.contiguous().view(B, S, H, 64).transpose(1, 2)
    ref = torch.nn.functional.scaled_dot_product_attention(heads(q).double(), heads(k).double(), heads(v).double())
    ref = ref.transpose(1, 2).reshape(B, S, i)
    lib = torch.nn.functional.scaled_dot_product_attention(heads(q), heads(k), heads(v)).transpose(1, 2).reshape(B, S, i)
    e = lambda t: (float((t.double() - ref).abs().max()), float((t.double() - ref).abs().mean()))
    print(f"input scale {scale_in}: kernel max / mean abs err {e(out)[0]:.3e} / {e(out)[1]:.3e}; library fp32 SDPA {e(lib)[0]:.3e} / {e(lib)[1]:.3e}; "
          f"|out| max {float(ref.abs().max()):.3f}")
