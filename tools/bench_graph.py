#!/usr/bin/env python3
"""Diagnostic: one 250-frame render step launched eagerly vs replayed from a captured HIP graph."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_motion_avatar_amd.config import RendererConfig  # noqa: E402
from audio_motion_avatar_amd.renderer import Renderer  # noqa: E402
from audio_motion_avatar_amd.synthetic import init_random_heads, make_render_inputs  # noqa: E402

F = 250
cfg = RendererConfig(image_size=(512, 512), subdivide_steps=0, predict_smplx_params=False, device="cuda")
r = init_random_heads(Renderer(cfg).eval())
tokens, smpl, cam = make_render_inputs(F, cfg, seed=42)
ws = [None]
with torch.no_grad():
    for _ in range(3):
        r.render_tokens(tokens[0], smpl, cam, workspaces=ws, check_overflow=True)
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        r.render_tokens(tokens[0], smpl, cam, workspaces=ws, check_overflow=False)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / n * 1e3
    t0 = time.perf_counter()
    for _ in range(n):
        r.render_tokens(tokens[0], smpl, cam, workspaces=ws, check_overflow=False)
    host = (time.perf_counter() - t0) / n * 1e3  # launch cost only (no sync)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.cuda.graph(g, stream=s):
        out, _ = r.render_tokens(tokens[0], smpl, cam, workspaces=ws, check_overflow=False)
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / n * 1e3
print(f"eager {eager:.3f} ms/step (host launch time {host:.3f} ms)   graph replay {graph:.3f} ms/step")
