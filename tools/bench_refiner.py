#!/usr/bin/env python3
"""Diagnostic: time of the point refiner (PTv3, reference configuration ptv3_encoder.yaml) per frame at the BASELINE
point count: LBS vertices -> triplane features -> PointTransformerV3 -> offsets.  Random weights, synthetic body.

    python tools/bench_refiner.py [frames] [clouds_per_pass] [num_points]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_motion_avatar_amd import ops  # noqa: E402
from audio_motion_avatar_amd.config import RendererConfig  # noqa: E402
from audio_motion_avatar_amd.renderer import Renderer  # noqa: E402
from audio_motion_avatar_amd.synthetic import init_random_heads, make_render_inputs  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
per_pass = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
cfg = RendererConfig(image_size=(512, 512), subdivide_steps=0, predict_smplx_params=False, no_point_refiner=False,
                     num_gaussians=N, refiner_clouds_per_pass=per_pass, device="cuda")
torch.manual_seed(0)
r = init_random_heads(Renderer(cfg).eval())
with torch.no_grad():
    r.point_refiner[-1].weight.normal_(0, 0.01)
tokens, smpl, cam = make_render_inputs(F, cfg, seed=42)
tok = tokens[0]
with torch.no_grad():
    verts = ops.points_gather(r._posed_vertices(smpl), r._gather_idx)
    for _ in range(2):
        out = r.refine_points(tok, verts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        out = r.refine_points(tok, verts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    pt = r.point_encoder.point_transformer
    pts = verts[:per_pass].contiguous()
    R = r._plane_resolution(tok)
    planes = tok.view(F, tok.shape[1], 3, R, R).permute(0, 2, 1, 3, 4)
    feats = ops.triplane_sample_features(planes[:per_pass], pts, cfg.radius)
    n = pts.shape[0] * pts.shape[1]
    cloud_of = torch.arange(pts.shape[0], device="cuda", dtype=torch.int32).repeat_interleave(pts.shape[1])
    grid, depth = ops.cloud_voxelize(pts.reshape(n, 3), cloud_of, pts.shape[0])
    print(f"frames {F}, {per_pass} per pass, {N} points: {dt * 1e3:.1f} ms = {dt / F * 1e3:.2f} ms / frame; "
          f"offset max {float((out - verts).abs().max()):.4f}; depth {depth.tolist()}")
    level = None
    import numpy as np
    from audio_motion_avatar_amd.point_transformer import Level
    level = Level(grid, cloud_of, depth, np.full(pts.shape[0], pts.shape[1]), ops.cloud_codes(grid, cloud_of, depth))
    sizes = [level.n]
    nb = level.neighbors(5)
    print(f"  level 0: {level.n} points, stem taps hit {float((nb >= 0).float().mean()) * 125:.1f} of 125")
    for s in range(1, pt.num_stages + 0):
        nb3 = level.neighbors(3)
        print(f"  level {s - 1}: {level.n} points, 3x3x3 taps hit {float((nb3 >= 0).float().mean()) * 27:.1f} of 27")
        level, _, _ = level.pool()
    nb3 = level.neighbors(3)
    print(f"  level {pt.num_stages - 1}: {level.n} points, 3x3x3 taps hit {float((nb3 >= 0).float().mean()) * 27:.1f} of 27; counts {level.counts.tolist()}")
