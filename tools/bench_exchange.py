#!/usr/bin/env python3
"""Diagnostic: stand-alone time of the exchange kernels for one 250-frame shard (pack with / without the rasterizer's
tile hint, dense uint8 pack, unpack of 1 and of 8 gathered shards)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_motion_avatar_amd import ops  # noqa: E402
from audio_motion_avatar_amd.config import RendererConfig  # noqa: E402
from audio_motion_avatar_amd.renderer import Renderer  # noqa: E402
from audio_motion_avatar_amd.synthetic import init_random_heads, make_render_inputs  # noqa: E402

F, H, W = 250, 512, 512
cfg = RendererConfig(image_size=(H, W), subdivide_steps=0, predict_smplx_params=False, device="cuda")
r = init_random_heads(Renderer(cfg).eval())
tokens, smpl, cam = make_render_inputs(F, cfg, seed=42)
ws = [None]
with torch.no_grad():
    rgba, _ = r.render_tokens(tokens[0], smpl, cam, workspaces=ws)
hint = ws[0].tile_counts()
count, _ = ops.frames_wire_count(ops.frames_pack_tiles(rgba, 0, tile_hint=hint))
cap = int(count * 1.25)
wire = ops.frames_pack_tiles(rgba, cap, tile_hint=hint)
print(f"stored tiles {count} of {F * 1024}; wire {wire.numel() / 1e6:.1f} MB, dense {F * H * W * 3 / 1e6:.1f} MB")


def timeit(name, fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"{name:34s} {(time.perf_counter() - t0) / n * 1e3:7.3f} ms")


dense = torch.empty(F, H, W, 3, dtype=torch.uint8, device="cuda")
timeit("dense uint8 pack", lambda: ops.frames_to_rgb8(rgba, out=dense))
timeit("sparse pack (pixel flags)", lambda: ops.frames_pack_tiles(rgba, cap, wire=wire))
timeit("sparse pack (rasterizer hint)", lambda: ops.frames_pack_tiles(rgba, cap, wire=wire, tile_hint=hint))
for nb in (1, 8):
    stack = wire[None].repeat(nb, 1)
    out = torch.empty(nb * F, H, W, 3, dtype=torch.uint8, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    timeit(f"unpack {nb} gathered shard(s)", lambda: ops.frames_unpack_tiles(stack, nb, F, H, W, cap, out=out, status=status))
    assert torch.equal(out[:F], ops.frames_to_rgb8(rgba)) and int(status.item()) == 0
    # differential unpack into the same (reused) buffer: steady state = the same tiles every step
    state = ops.frames_tile_state(nb, F, H, W, "cuda")
    out.random_(0, 255)
    timeit(f"delta unpack {nb} shard(s), steady", lambda: ops.frames_unpack_tiles(stack, nb, F, H, W, cap, out=out, status=status, state=state))
    assert torch.equal(out[:F], ops.frames_to_rgb8(rgba)) and int(status.item()) == 0
    # worst case for the differential form: every step the body sits elsewhere (alternate two different shards)
    rgba2 = torch.roll(rgba, shifts=(64, 96), dims=(1, 2)).contiguous()
    wire2 = ops.frames_pack_tiles(rgba2, cap)
    stack2 = wire2[None].repeat(nb, 1)
    flip = [0]

    def alternate():
        flip[0] ^= 1
        ops.frames_unpack_tiles(stack2 if flip[0] else stack, nb, F, H, W, cap, out=out, status=status, state=state)

    timeit(f"delta unpack {nb} shard(s), moving", alternate)
    ops.frames_unpack_tiles(stack, nb, F, H, W, cap, out=out, status=status, state=state)
    assert torch.equal(out[:F], ops.frames_to_rgb8(rgba))
