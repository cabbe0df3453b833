#!/usr/bin/env python3
"""Diagnostic: the library convolutions on a batch whose activations exceed 4 GiB (18 planes x 256 x 512 x 512 fp32 =
4.8 GB, what TriplaneUpsampler.forward sees for a 6-frame window) against the same planes in chunks."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace  # noqa: E402

from audio_motion_avatar_amd.renderer import TriplaneUpsampler  # noqa: E402

torch.manual_seed(0)
up = TriplaneUpsampler(SimpleNamespace(triplane_feature_dim=256, num_upsample_blocks=4)).eval().cuda()
x = torch.randn(18, 256, 32, 32, device="cuda")
with torch.no_grad():
    whole, _ = up._run(x)
    for i in range(0, 18, 3):
        part, _ = up._run(x[i:i + 3])
        err = (whole[i:i + 3] - part).abs().amax(dim=(1, 2, 3)).tolist()
        print(f"items {i}..{i + 2} (bytes {i * 268435456 / 2**30:.2f} GiB ..): max |whole batch - chunk| {['%.1e' % e for e in err]}", flush=True)
