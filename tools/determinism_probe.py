#!/usr/bin/env python3
"""Diagnostic: is one transformer step bit-reproducible from call to call (tuned library GEMMs on / off)?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_motion_avatar_amd.transformer import Transformer1D_nn  # noqa: E402
from audio_motion_avatar_amd.tuning import use_tuned_gemms  # noqa: E402

print("tuned GEMM lookup:", use_tuned_gemms())
torch.manual_seed(0)
net = Transformer1D_nn(num_layers=8, attention_head_dim=64, in_channels=256, num_attention_heads=8,
                       cross_attention_dim=768).cuda().eval()
for B in (1, 2):
    x = torch.randn(B, 256, 6304, device="cuda")
    a = torch.randn(B, 1, 768, device="cuda")
    with torch.no_grad():
        ref = net(x, a)
        same = [bool(torch.equal(net(x, a), ref)) for _ in range(6)]
    print(f"B={B}: repeated calls bit-identical: {same}")
