#!/usr/bin/env python3
"""Diagnostic: standalone time of the LBS launch (joint chain + skin) for 250 frames."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_motion_avatar_amd import ops  # noqa: E402
from audio_motion_avatar_amd.body_model import BodyModel  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 250
body = BodyModel.synthetic_model(42, "cuda")
pose = torch.randn(F, 165, device="cuda") * 0.2
coef = torch.randn(F, 20, device="cuda")
for _ in range(3):
    ops.lbs_forward(body.device_tables(), pose, coef)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    ops.lbs_forward(body.device_tables(), pose, coef)
torch.cuda.synchronize()
print(f"AMAV_LBS_FT={os.environ.get('AMAV_LBS_FT', 'default')}: lbs_forward F={F}: {(time.perf_counter() - t0) / n * 1e6:.1f} us")
