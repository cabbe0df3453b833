#!/usr/bin/env python3
"""Diagnostic: block-level phase stamps of the binning kernel (library built with -DAMAV_BIN_STAMPS, tools/stamp_bin.sh)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_motion_avatar_amd import ops  # noqa: E402
from audio_motion_avatar_amd.config import RendererConfig  # noqa: E402
from audio_motion_avatar_amd.renderer import Renderer, render_batch  # noqa: E402
from audio_motion_avatar_amd.synthetic import init_random_heads, make_render_inputs  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 250
cfg = RendererConfig(image_size=(512, 512), subdivide_steps=0, predict_smplx_params=False, device="cuda")
r = init_random_heads(Renderer(cfg).eval())
tokens, smpl, cam = make_render_inputs(F, cfg, seed=42)
T = 32 * 32
with torch.no_grad():
    pts = r.get_smpl_vertices(smpl)
    g = r.unpack_gaussians(r.decode_gaussians(tokens[0], pts, smpl["transl"].reshape(F, 3)))
    for _ in range(2):
        render_batch(g, cam["intrinsic"], cam["extrinsic"], cfg)
    stamps = torch.zeros(F * T * 6 + F * 8, dtype=torch.int64, device="cuda")
    ops.DEBUG_STAMPS = stamps
    render_batch(g, cam["intrinsic"], cam["extrinsic"], cfg)
    torch.cuda.synchronize()
    ops.DEBUG_STAMPS = None
b = stamps[F * T * 6:].cpu().numpy().astype(np.int64).reshape(F, 8)
US = 0.01
t0 = b[:, 0].min()
print(f"blocks start {((b[:, 0] - t0) * US).min():.1f}..{((b[:, 0] - t0) * US).max():.1f} us, last block ends {((b[:, 4] - t0) * US).max():.1f} us")
names = ["project + count", "scan + offsets", "wire slots + queues", "key scatter"]
for k, name in enumerate(names):
    d = (b[:, k + 1] - b[:, k]) * US
    print(f"  {name:22s} mean {d.mean():6.1f} us  min {d.min():6.1f}  max {d.max():6.1f}")
