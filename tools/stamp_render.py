#!/usr/bin/env python3
"""Diagnostic: per-phase clock stamps of the blend kernel on the bench workload (not part of the product path)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_motion_avatar_amd import ops  # noqa: E402
from audio_motion_avatar_amd.config import RendererConfig  # noqa: E402
from audio_motion_avatar_amd.renderer import Renderer, render_batch  # noqa: E402
from audio_motion_avatar_amd.synthetic import init_random_heads, make_render_inputs  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 250
DETAIL = "--detail" in sys.argv  # the library was built with -DAMAV_STAMP_DETAIL (tools/stamp_detail.sh)
cfg = RendererConfig(image_size=(512, 512), subdivide_steps=0, predict_smplx_params=False, device="cuda")
r = init_random_heads(Renderer(cfg).eval())
tokens, smpl, cam = make_render_inputs(F, cfg, seed=42)
T = 32 * 32
with torch.no_grad():
    pts = r.get_smpl_vertices(smpl)
    g = r.unpack_gaussians(r.decode_gaussians(tokens[0], pts, smpl["transl"].reshape(F, 3)))
    for _ in range(2):
        render_batch(g, cam["intrinsic"], cam["extrinsic"], cfg)
    stamps = torch.zeros(F * T, 6, dtype=torch.int64, device="cuda")
    ops.DEBUG_STAMPS = stamps
    render_batch(g, cam["intrinsic"], cam["extrinsic"], cfg)
    torch.cuda.synchronize()
    ops.DEBUG_STAMPS = None
s = stamps.cpu().numpy().astype(np.int64)
n = s[:, 5]
ne = s[:, 0] != 0
US = 0.01  # s_memrealtime ticks at 100 MHz
t0 = s[ne, 0].min()
span = (s[ne, 4].max() - t0) * US
print(f"tiles {len(n)}, nonempty {ne.sum()}, mean n (nonempty) {n[ne].mean():.1f}, max n {n.max()}, kernel span {span:.1f} us")
sn = s[ne]
if DETAIL:
    tile = (sn[:, 4] - sn[:, 0]) * US
    asm_t = sn[:, 1] * US
    prep_t = (sn[:, 2] >> 48) * US
    fill_t = ((sn[:, 2] >> 32) & 0xffff) * US
    stage_t = ((sn[:, 2] >> 16) & 0xffff) * US
    own_t = (sn[:, 2] & 0xffff) * US
    store_t = (sn[:, 4] - sn[:, 3]) * US
    rest = tile - asm_t - prep_t - fill_t - store_t - stage_t - own_t
    waves = 4096
    print(f"per wave (/{waves}): in tiles {tile.sum() / waves:7.1f} us = blend loops {asm_t.sum() / waves:6.1f} + next tile's sort and gather issue "
          f"{prep_t.sum() / waves:6.1f} + background stores {fill_t.sum() / waves:6.1f} + tile store {store_t.sum() / waves:6.1f} + "
          f"staging (wait for records, masks, lists) {stage_t.sum() / waves:6.1f} + own sort (unprepared tiles) {own_t.sum() / waves:6.1f} + "
          f"rest {rest.sum() / waves:6.1f}; between tiles {span - tile.sum() / waves:6.1f}")
    nn = sn[:, 5]
    for lo, hi in ((1, 32), (32, 64), (64, 128), (128, 256), (256, 512), (512, 10**9)):
        m = (nn >= lo) & (nn < hi)
        if m.any():
            print(f"  n in [{lo},{hi}): {m.sum():6d} tiles  tile {tile[m].mean():7.2f} us  blend {asm_t[m].mean():7.2f}  prep {prep_t[m].mean():6.2f}  "
                  f"fill {fill_t[m].mean():6.2f}  store {store_t[m].mean():5.2f}  stage {stage_t[m].mean():6.2f}  own {own_t[m].mean():6.2f}  rest {rest[m].mean():7.2f}   blend/gaussian {asm_t[m].sum() / nn[m].sum() * 1e3:6.1f} ns")
    sys.exit(0)
ph = [(sn[:, i + 1] - sn[:, i]) * US for i in range(4)]
for nm, d in zip(["read ranges", "load+sort", "blend", "store"], ph):
    print(f"  {nm:12s} wave-time total {d.sum() / 1e3:8.2f} ms   mean {d.mean():7.2f} us   p99 {np.percentile(d, 99):7.2f} us")
for lo, hi in ((1, 32), (32, 64), (64, 128), (128, 256), (256, 512), (512, 10**9)):
    m = (sn[:, 5] >= lo) & (sn[:, 5] < hi)
    if m.any():
        print(f"  n in [{lo},{hi}): {m.sum():6d} tiles  sort {ph[1][m].mean():7.2f} us  blend {ph[2][m].mean():7.2f} us  "
              f"blend/gaussian {ph[2][m].sum() / sn[m, 5].sum() * 1e3:6.1f} ns")
ev = np.concatenate([np.stack([sn[:, 0], np.ones(len(sn))], 1), np.stack([sn[:, 4], -np.ones(len(sn))], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
conc = np.cumsum(ev[:, 1])
ts = (ev[:, 0] - t0) * US
for q in np.linspace(0, ts.max(), 21)[1:]:
    i = np.searchsorted(ts, q) - 1
    hv = ((sn[:, 0] - t0) * US <= q) & ((sn[:, 4] - t0) * US >= q)
    print(f"  t={q:7.1f} us  waves in flight {int(conc[i]):5d}  (nonempty {int(hv.sum())})")
# who is still running at the end?
end = (sn[:, 4] - t0) * US
start = (sn[:, 0] - t0) * US
late = end > 0.9 * ts.max()
print(f"tiles finishing in the last 10 % of the kernel: {late.sum()}")
for lo, hi in ((1, 32), (32, 64), (64, 128), (128, 256), (256, 512), (512, 10**9)):
    m = late & (sn[:, 5] >= lo) & (sn[:, 5] < hi)
    if m.any():
        print(f"  n in [{lo},{hi}): {m.sum():5d} tiles  start {start[m].min():6.1f}..{start[m].max():6.1f} us  "
              f"duration mean {(end[m] - start[m]).mean():6.1f} max {(end[m] - start[m]).max():6.1f} us")
# dispatch order: when does each length class start?
for lo, hi in ((1, 32), (32, 64), (64, 128), (128, 256), (256, 512), (512, 10**9)):
    m = (sn[:, 5] >= lo) & (sn[:, 5] < hi)
    if m.any():
        print(f"  class [{lo},{hi}): starts {np.percentile(start[m], 1):6.1f} .. {np.percentile(start[m], 99):6.1f} us (p1..p99)")
# per work queue (= XCD): queue of a tile = (band of its tile row + frame) % 8 (bin_kernel); when does each queue's last
# tile end, and how much tile time did it carry?
gy = gx = 32
idx = np.nonzero(ne)[0]
fr, tl = idx // T, idx % T
band = np.minimum(7, (tl // gx) * 8 // gy)
qu = (band + fr) % 8
for qq in range(8):
    m = qu == qq
    e = sn[m, 4]
    print(f"  queue {qq}: {m.sum():6d} tiles, instances {n[ne][m].sum():8d}, tile wave-time {((sn[m, 4] - sn[m, 0]) * US).sum() / 1e3:7.2f} ms, "
          f"last tile ends {(e.max() - t0) * US:6.1f} us, 99 % of its tiles ended by {(np.percentile(e, 99) - t0) * US:6.1f} us")
