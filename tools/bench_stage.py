#!/usr/bin/env python3
"""Diagnostic: standalone times of the pre-raster stages on the bench workload (project, decode, LBS)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_motion_avatar_amd import ops  # noqa: E402
from audio_motion_avatar_amd.config import RendererConfig  # noqa: E402
from audio_motion_avatar_amd.renderer import Renderer  # noqa: E402
from audio_motion_avatar_amd.synthetic import init_random_heads, make_render_inputs  # noqa: E402

F = 250
cfg = RendererConfig(image_size=(512, 512), subdivide_steps=0, predict_smplx_params=False, device="cuda",
                     subset_order=os.environ.get("AMAV_SUBSET_ORDER", RendererConfig.subset_order))
r = init_random_heads(Renderer(cfg).eval())
tokens, smpl, cam = make_render_inputs(F, cfg, seed=42)
w_plane, w_point = r._head_weights()


def timeit(name, fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"{name:28s} {(time.perf_counter() - t0) / n * 1e6:8.1f} us")


with torch.no_grad():
    proj = ops.triplane_project(tokens[0], w_plane, 32)
    verts = r._posed_vertices(smpl)
    transl = smpl["transl"].reshape(F, 3)
    timeit("triplane_project", lambda: ops.triplane_project(tokens[0], w_plane, 32))
    timeit("lbs (chain+skin)", lambda: r._posed_vertices(smpl))
    timeit("sample_decode_indexed", lambda: ops.triplane_sample_decode_indexed(proj, verts, r._gather_idx, transl, 1.4, w_point))
    timeit("gaussians_from_tokens", lambda: r.gaussians_from_tokens(tokens[0], smpl))
    x = tokens[0]
    timeit("torch clone 786 MB (r+w)", lambda: x.clone())
    timeit("torch sum 786 MB (read)", lambda: x.sum())
