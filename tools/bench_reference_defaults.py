#!/usr/bin/env python3
"""Diagnostic: one 6-frame window through Renderer.forward with the reference's DEFAULT renderer.yaml (triplane
upsampler x16, PTv3 point refiner, subdivide_steps = 2 -> 30 000 Gaussians, 512 x 512) -- the configuration a
released checkpoint would run with; BASELINE's configs name neither the upsampler nor the refiner.

    python tools/bench_reference_defaults.py [frames]
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

F = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for label, kw in (("upsampler + refiner, 30k", dict(upsample_triplane=True, no_point_refiner=False, subdivide_steps=2)),
                  ("same, margin 0", dict(upsample_triplane=True, no_point_refiner=False, subdivide_steps=2, upsample_window_margin=0.0)),
                  ("upsampler only, 30k", dict(upsample_triplane=True, subdivide_steps=2)),
                  ("refiner only, 30k", dict(no_point_refiner=False, subdivide_steps=2)),
                  ("neither, 30k", dict(subdivide_steps=2)), ("neither, 10k (BASELINE)", dict(subdivide_steps=0))):
    cfg = RendererConfig(image_size=(512, 512), predict_smplx_params=False, device="cuda", **kw)
    torch.manual_seed(0)
    r = init_random_heads(Renderer(cfg).eval())
    if hasattr(r, "point_refiner"):
        with torch.no_grad():
            r.point_refiner[-1].weight.normal_(0, 0.005)
    tokens, smpl, cam = make_render_inputs(F, cfg, seed=42)
    dummy = torch.zeros(1, F, 1, 1, device="cuda")
    with torch.no_grad():
        for _ in range(2):
            r(tokens, cam, dummy, smpl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            images, _ = r(tokens, cam, dummy, smpl)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3 / F
    plan = getattr(r, "last_window_plan", None)
    tiles = [round(int(w["mask"].sum()) / F, 1) for w in plan] if plan else None
    print(f"{label:28s} {dt * 1e3:8.2f} ms per frame ({F} frames per call), coverage {float((images < 0.999).any(-1).float().mean()):.3f}"
          f", active tiles of 64 per plane {tiles}", flush=True)
    del r
    torch.cuda.empty_cache()
