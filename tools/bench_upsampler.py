#!/usr/bin/env python3
"""Diagnostic: time of the TriplaneUpsampler (renderer.py:377-417) at the reference defaults (4 blocks, C=256,
32^2 -> 512^2, three planes per frame) and of the default-config frame (upsampler + refiner at 30 000 points).

    python tools/bench_upsampler.py [frames]
"""
import os
import sys
import time

import torch

if os.environ.get("AMAV_CONV_BENCHMARK"):
    torch.backends.cudnn.benchmark = True  # MIOpen find mode: search the convolution kernels per shape

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_motion_avatar_amd.config import RendererConfig  # noqa: E402
from audio_motion_avatar_amd.renderer import Renderer  # noqa: E402
from audio_motion_avatar_amd.synthetic import init_random_heads, make_render_inputs  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = RendererConfig(image_size=(512, 512), subdivide_steps=0, predict_smplx_params=False, upsample_triplane=True,
                     num_upsample_blocks=4, device="cuda")
r = init_random_heads(Renderer(cfg).eval())
tokens, smpl, cam = make_render_inputs(F, cfg, seed=42)
up = r.triplane_upsampler


def flops():
    c, res, total = cfg.triplane_feature_dim, cfg.triplane_resolution, 0.0
    for i in range(cfg.num_upsample_blocks):
        res *= 2
        total += 3 * (2.0 * res * res * c * c * 9) + (2.0 * (res // 2) ** 2 * c * c if i == 0 else 0.0)
    return 3 * total  # three planes


with torch.no_grad():
    for _ in range(2):
        out = up.forward_tokens(tokens[0], cfg.triplane_resolution)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        out = up.forward_tokens(tokens[0], cfg.triplane_resolution)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps / F
print(f"TriplaneUpsampler, full planes: {dt * 1e3:.1f} ms per frame, {flops() / 1e12:.2f} TFLOP per frame -> "
      f"{flops() / dt / 1e12:.1f} TFLOP/s; output {tuple(out.shape)}")

# windowed: blocks 1..3 on the bounding box of the active tiles, the last block on the active tiles only
with torch.no_grad():
    pts = r.get_smpl_vertices(smpl)
    plan = up.plan_windows(pts, cfg.triplane_resolution, cfg.radius)
    for _ in range(2):
        win = up.forward_tokens_windowed(tokens[0], cfg.triplane_resolution, plan)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan = up.plan_windows(pts, cfg.triplane_resolution, cfg.radius)
        win = up.forward_tokens_windowed(tokens[0], cfg.triplane_resolution, plan)
    torch.cuda.synchronize()
    dw = (time.perf_counter() - t0) / reps / F
    R = cfg.triplane_resolution
    cells = sum((w["crop"][1] - w["crop"][0]) * (w["crop"][3] - w["crop"][2]) for w in plan) / (3 * R * R)
    tiles = sum(int(w["mask"].sum()) for w in plan) / (3 * F * (R // 4) ** 2)
    r_out, t = R * 16, 64
    fv, wv = out.view(F, -1, 3, r_out, r_out), win.view(F, -1, 3, r_out, r_out)
    err = max(float((fv[f, :, p, ty * t:(ty + 1) * t, tx * t:(tx + 1) * t] - wv[f, :, p, ty * t:(ty + 1) * t, tx * t:(tx + 1) * t]).abs().max())
              for p, w in enumerate(plan) for f, ty, tx in torch.nonzero(w["mask"]).tolist())
print(f"windowed: {dw * 1e3:.1f} ms per frame; crops {[w['crop'] for w in plan]} = {cells * 100:.0f} % of the cells for blocks 1-3, "
      f"{tiles * 100:.0f} % of the tiles for block 4 (tiled: {[w['tiles'] is not None for w in plan]}); "
      f"max |full - windowed| inside the active tiles {err:.2e}")
