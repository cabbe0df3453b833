#!/usr/bin/env python3
"""Diagnostic: where a wave of selfattn_f16_kernel spends its clocks, per 64-key tile (reference shape S=6304, H=8).
Needs a stamped build of the library (not the product build):
    cd audio-motion-avatar_amd/csrc && touch attention.hip && make CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -DAMAV_ATTN_STAMPS"
    python tools/attention_stamps.py          (then rebuild with plain `make`)
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_motion_avatar_amd import _lib, ops  # noqa: E402

S, H = 6304, 8
q, k, v = (torch.randn(1, S, H * 64, device="cuda") for _ in range(3))
lib = _lib.lib()
if not hasattr(lib, "amav_debug_attn_stamps"):
    raise SystemExit("this libamav_hip.so was built without -DAMAV_ATTN_STAMPS (see the docstring)")
buf = (ctypes.c_ulonglong * 8)()
ops.selfattn(q, k, v, H)
lib.amav_debug_attn_stamps(buf)  # clear the warm-up
calls = 5
for _ in range(calls):
    ops.selfattn(q, k, v, H)
lib.amav_debug_attn_stamps(buf)
waves = buf[0]
tiles = calls * H * ((S + 31) // 32) * ((S + 63) // 64)  # (32-query wave, 64-key tile) pairs
names = ["", "stage + barrier", "QK^T (24 MFMA)", "max, correction, first split", "PV (24 MFMA) + pipelined splits",
         "second barrier"]
total = sum(buf[1:6])
print(f"{waves} waves, {tiles} wave-tiles, {total / tiles:.0f} clocks per wave-tile (MFMA alone: 48 x 32 = 1536)")
for i in range(1, 6):
    print(f"  {names[i]:34s} {buf[i] / tiles:8.0f} clocks  {100.0 * buf[i] / total:5.1f} %")
