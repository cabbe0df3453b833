#!/usr/bin/env python3
"""Diagnostic: the rasterizer's launches at the reference's own window (6 frames, 30 000 Gaussians) and at the bench
shard (250 frames, 10 000 Gaussians); run under `rocprofv3 --kernel-trace` and summarise with tools/kernel_stats.py.

    python tools/bench_binning.py [frames gaussians_level]      level: 0 -> 10 000, 1 -> 30 000 Gaussians
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_motion_avatar_amd.config import RendererConfig  # noqa: E402
from audio_motion_avatar_amd.renderer import Renderer  # noqa: E402
from audio_motion_avatar_amd.synthetic import init_random_heads, make_render_inputs  # noqa: E402

for F, steps in ((6, 2), (250, 0)) if len(sys.argv) < 3 else ((int(sys.argv[1]), int(sys.argv[2])),):
    cfg = RendererConfig(image_size=(512, 512), subdivide_steps=steps, predict_smplx_params=False, device="cuda")
    r = init_random_heads(Renderer(cfg).eval())
    tokens, smpl, cam = make_render_inputs(F, cfg, seed=42)
    ws = [None]
    with torch.no_grad():
        for _ in range(12):
            r.render_tokens(tokens[0], smpl, cam, workspaces=ws)
    torch.cuda.synchronize()
    print(f"F={F} N={r.num_verts}: done", flush=True)
