"""Size-independent properties of the HIP path at BASELINE.json's full sizes (250 frames x 512x512 x 10 000 Gaussians,
SMPL-X sized body), where the CPU oracle would take minutes: instead of a second implementation these tests use
identities the algorithms satisfy exactly.

  rasterizer   a frame's pixels do not depend on the other frames of the launch, nor on the order the Gaussians are
               stored in (the blend order is the (depth, index) sort; distinct depths => same order), the output is
               deterministic run to run (atomics only decide where a key lands before it is sorted), alpha and inverse
               depth do not depend on the colours, and RGB is linear in (colours, background)
  LBS          a global rotation turns the posed mesh rigidly about the pelvis
  decode       the root translation only shifts xyz
  exchange     the uint8 packing equals the reference's (frame * 255).astype(uint8)
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

F, N, H, W = 250, 10000, 512, 512


@pytest.fixture(scope="module")
def full_clip():
    """Gaussians of the bench workload (configs[1]) decoded by the HIP path, plus cameras."""
    from audio_motion_avatar_amd import ops
    from audio_motion_avatar_amd.config import RendererConfig
    from audio_motion_avatar_amd.renderer import Renderer
    from audio_motion_avatar_amd.synthetic import init_random_heads, make_render_inputs

    cfg = RendererConfig(image_size=(H, W), subdivide_steps=0, predict_smplx_params=False, device="cuda")
    r = init_random_heads(Renderer(cfg).eval())
    tokens, smpl, cam = make_render_inputs(F, cfg, seed=42, device="cuda")
    with torch.no_grad():
        packed = r.gaussians_from_tokens(tokens[0], smpl)
    g = {k: v.contiguous() for k, v in r.unpack_gaussians(packed).items() if k != "shs"}
    view, proj, tanfov, _ = ops.camera_from_intrinsics(cam["intrinsic"][0].float(), cam["extrinsic"][0].float(), H, W)
    assert g["xyz"].shape == (F, N, 3)
    return dict(g=g, view=view, proj=proj, tanfov=tanfov, renderer=r, tokens=tokens, smpl=smpl, cam=cam)


def raster(c, sel=slice(None), perm=None, color=None, bg=(1.0, 1.0, 1.0), **kw):
    from audio_motion_avatar_amd import ops

    g = {k: v[sel] for k, v in c["g"].items()}
    if color is not None:
        g["color"] = color[sel]
    if perm is not None:
        g = {k: v[:, perm].contiguous() for k, v in g.items()}
    out = ops.rasterize(g["xyz"], g["rot"], g["scale"], g["opacity"], g["color"], c["view"][sel], c["proj"][sel],
                        c["tanfov"][sel], H, W, bg=bg, apply_activations=True, **kw)
    assert not out["workspace"].status()[1]
    return out


def test_rasterizer_frames_are_independent_and_deterministic(full_clip):
    whole = raster(full_clip, clamp_output=True)["rgba"]
    again = raster(full_clip, clamp_output=True)["rgba"]
    assert torch.equal(whole, again)
    part = raster(full_clip, sel=slice(100, 117), clamp_output=True)["rgba"]
    assert torch.equal(whole[100:117], part)
    cover = (whole[..., 3] > 0.5).float().mean().item()
    assert 0.03 < cover < 0.6, f"degenerate workload: coverage {cover}"


def test_rasterizer_ignores_storage_order(full_clip):
    sel = slice(40, 56)
    base = raster(full_clip, sel=sel, want_inv_depth=True, want_radii=True)
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(3)).cuda()
    shuf = raster(full_clip, sel=sel, perm=perm, want_inv_depth=True, want_radii=True)
    assert torch.equal(base["radii"][:, perm], shuf["radii"])
    assert base["workspace"].status()[0] == shuf["workspace"].status()[0]  # same instance count
    # Gaussians of exactly equal view depth are blended in index order (upstream's stable sort), i.e. in storage
    # order: the few pixels under such a pair may swap the two; every other pixel is bit-identical
    same = (base["rgba"] == shuf["rgba"]).all(-1)
    assert same.float().mean().item() > 0.9999
    assert (base["rgba"] - shuf["rgba"]).abs().max().item() <= 1e-3
    assert (base["inv_depth"] - shuf["inv_depth"]).abs().max().item() <= 1e-3


def test_rasterizer_linear_in_colour_and_background(full_clip):
    sel = slice(200, 216)
    gen = torch.Generator().manual_seed(11)
    c1 = torch.rand(F, N, 3, generator=gen).cuda()
    c2 = torch.rand(F, N, 3, generator=gen).cuda()
    # logits: the kernel applies no activation to colours (clamp to [0,1] is the caller's, renderer.py:546-547)
    b1, b2 = (0.2, 0.9, 0.4), (1.0, 0.0, 0.5)
    a, b = 0.25, 0.75
    o1 = raster(full_clip, sel=sel, color=c1, bg=b1, want_inv_depth=True)
    o2 = raster(full_clip, sel=sel, color=c2, bg=b2, want_inv_depth=True)
    mix = raster(full_clip, sel=sel, color=a * c1 + b * c2, bg=tuple(a * x + b * y for x, y in zip(b1, b2)),
                 want_inv_depth=True)
    # alpha and inverse depth never see the colours
    assert torch.equal(o1["rgba"][..., 3], o2["rgba"][..., 3]) and torch.equal(o1["rgba"][..., 3], mix["rgba"][..., 3])
    assert torch.equal(o1["inv_depth"], o2["inv_depth"])
    lin = a * o1["rgba"][..., :3] + b * o2["rgba"][..., :3]
    assert (mix["rgba"][..., :3] - lin).abs().max().item() <= 5e-6


def test_lbs_global_rotation_is_rigid_about_the_pelvis(full_clip):
    from audio_motion_avatar_amd import ops

    body = full_clip["renderer"].smplx_model
    tables = body.device_tables()
    gen = torch.Generator().manual_seed(5)
    pose = (torch.randn(F, 165, generator=gen) * 0.25).cuda()
    coeffs = torch.randn(F, 20, generator=gen).cuda()
    pose0 = pose.clone()
    pose0[:, :3] = 0
    v0 = ops.lbs_forward(tables, pose0, coeffs)
    v1 = ops.lbs_forward(tables, pose, coeffs)
    # Rodrigues of the global orientation in fp64 (the kernel's eps only matters at zero angle)
    r = pose[:, :3].double()
    ang = r.norm(dim=1, keepdim=True)
    k = r / ang
    Kx = torch.zeros(F, 3, 3, dtype=torch.float64, device="cuda")
    Kx[:, 0, 1], Kx[:, 0, 2], Kx[:, 1, 0] = -k[:, 2], k[:, 1], k[:, 2]
    Kx[:, 1, 2], Kx[:, 2, 0], Kx[:, 2, 1] = -k[:, 0], -k[:, 1], k[:, 0]
    R = torch.eye(3, dtype=torch.float64, device="cuda") + torch.sin(ang)[..., None] * Kx \
        + (1 - torch.cos(ang))[..., None] * (Kx @ Kx)
    J0 = (tables["j_template"][0].double() + (tables["j_dirs"][0:3].double() @ coeffs.double().T).T)  # pelvis [F,3]
    expect = torch.einsum("fij,fvj->fvi", R, v0.double() - J0[:, None]) + J0[:, None]
    assert (v1.double() - expect).abs().max().item() <= 1e-5


def test_decode_translation_only_shifts_xyz(full_clip):
    r, tokens, smpl = full_clip["renderer"], full_clip["tokens"], full_clip["smpl"]
    moved = dict(smpl)
    shift = torch.tensor([0.125, -0.25, 0.5], device="cuda")  # exactly representable: the sums below round alike
    moved["transl"] = smpl["transl"] + shift
    with torch.no_grad():
        a = r.gaussians_from_tokens(tokens[0], smpl).clone()
        b = r.gaussians_from_tokens(tokens[0], moved)
    ga, gb = r.unpack_gaussians(a), r.unpack_gaussians(b)
    for k in ("rot", "scale", "opacity", "color"):
        assert torch.equal(ga[k], gb[k]), k
    assert (gb["xyz"] - ga["xyz"] - shift).abs().max().item() <= 1e-6


def test_rgb8_packing_matches_the_reference_quantisation(full_clip):
    from audio_motion_avatar_amd import ops

    rgba = raster(full_clip, sel=slice(0, 32), clamp_output=True)["rgba"]
    got = ops.frames_to_rgb8(rgba)
    want = (rgba[..., :3] * 255).to(torch.uint8)  # src/main2.py:351
    assert torch.equal(got, want)
    assert math.isclose(float(got.float().mean()), float(want.float().mean()))


def test_sparse_wire_round_trip_is_lossless(full_clip):
    """pack -> (gather of two different shards, emulated by concatenation) -> unpack == dense uint8 frames, and a wire
    that is too small is reported instead of silently dropping tiles."""
    from audio_motion_avatar_amd import ops

    rgba = raster(full_clip, sel=slice(0, 48), clamp_output=True)["rgba"]
    a, b = rgba[:24].contiguous(), rgba[24:].contiguous()
    count_a, _ = ops.frames_wire_count(ops.frames_pack_tiles(a, 0))
    count_b, _ = ops.frames_wire_count(ops.frames_pack_tiles(b, 0))
    tiles = 24 * 32 * 32
    assert 0 < count_a < tiles // 2 and 0 < count_b < tiles // 2  # an avatar clip is mostly background
    cap = max(count_a, count_b) + 7
    wa, wb = ops.frames_pack_tiles(a, cap), ops.frames_pack_tiles(b, cap)
    assert ops.frames_wire_count(wa) == (count_a, cap)
    assert wa.numel() == ops.frames_wire_bytes(24, H, W, cap) < 24 * H * W * 3 // 2
    out, status = ops.frames_unpack_tiles(torch.stack([wa, wb]), 2, 24, H, W, cap)
    assert torch.equal(out, ops.frames_to_rgb8(rgba))
    assert int(status.item()) == 0
    # too small: flagged
    small = max(count_a, count_b) - 5
    ws_ = torch.stack([ops.frames_pack_tiles(a, small), ops.frames_pack_tiles(b, small)])
    _, status = ops.frames_unpack_tiles(ws_, 2, 24, H, W, small)
    assert int(status.item()) == 1


def test_sparse_wire_handles_partial_tiles_and_other_backgrounds():
    from audio_motion_avatar_amd import ops

    g = torch.Generator().manual_seed(2)
    F_, H_, W_ = 3, 50, 70  # neither a multiple of 16; W not a multiple of 4 -> byte stores
    for W2 in (W_, 72):
        rgba = torch.zeros(F_, H_, W2, 4)
        rgba[..., 0], rgba[..., 1], rgba[..., 2] = 0.25, 0.5, 0.75  # background
        rgba[1, 10:30, 5:40, :3] = torch.rand(20, 35, 3, generator=g)
        rgba[2, 45:, 60:, :3] = torch.rand(5, W2 - 60, 3, generator=g)
        rgba = rgba.cuda().contiguous()
        bg = (0.25, 0.5, 0.75)
        count, _ = ops.frames_wire_count(ops.frames_pack_tiles(rgba, 0, bg))
        assert 0 < count <= 3 * 3 + 1  # the two patches touch a handful of tiles; frame 0 none
        wire = ops.frames_pack_tiles(rgba, count, bg)
        out, status = ops.frames_unpack_tiles(wire[None], 1, F_, H_, W2, count)
        want = (rgba[..., :3].clamp(0, 1) * 255).to(torch.uint8)
        assert torch.equal(out, want) and int(status.item()) == 0


def test_sparse_wire_with_the_rasterizer_tile_hint(full_clip):
    """The rasterizer's per-tile list lengths as the "may differ from background" hint: a superset of the tiles the
    pixel test stores, same frames after unpacking, no pass over the fp32 frames for the flags."""
    from audio_motion_avatar_amd import ops

    out = raster(full_clip, sel=slice(60, 92), clamp_output=True)
    rgba, hint = out["rgba"], out["workspace"].tile_counts()
    assert hint.shape == (32 * 32 * 32,) and int((hint > 0).sum()) > 0
    exact, _ = ops.frames_wire_count(ops.frames_pack_tiles(rgba, 0))
    hinted, _ = ops.frames_wire_count(ops.frames_pack_tiles(rgba, 0, tile_hint=hint))
    assert exact <= hinted == int((hint > 0).sum()) <= 1.2 * exact + 8
    wire = ops.frames_pack_tiles(rgba, hinted, tile_hint=hint)
    dense, status = ops.frames_unpack_tiles(wire[None], 1, 32, H, W, hinted)
    assert torch.equal(dense, ops.frames_to_rgb8(rgba)) and int(status.item()) == 0


def test_rasterizer_writes_the_wire_format_itself(full_clip):
    """with amav_raster_args.wire the blend kernel's write-back also emits the exchange's tile-sparse
    wire buffer (slots handed out by the binning kernel): after unpacking it equals the uint8 frames, its count and its
    stored-tile set equal the pack pass with the tile counts as hint, and a capacity that is too small drops tiles and
    says so -- with another background colour and with a frame size that has partial edge tiles as well."""
    from audio_motion_avatar_amd import ops

    sel, Fs = slice(60, 92), 32
    tiles = Fs * 32 * 32
    cap = tiles // 3
    wire = torch.zeros(ops.frames_wire_bytes(Fs, H, W, cap), dtype=torch.uint8, device="cuda")
    bg = (0.2, 0.5, 0.9)
    out = raster(full_clip, sel=sel, clamp_output=True, bg=bg, wire=(wire, cap))
    rgba, hint = out["rgba"], out["workspace"].tile_counts()
    count, capacity = ops.frames_wire_count(wire)
    assert capacity == cap and count == int((hint > 0).sum()) <= cap
    dense, status = ops.frames_unpack_tiles(wire[None], 1, Fs, H, W, cap)
    assert torch.equal(dense, ops.frames_to_rgb8(rgba)) and int(status.item()) == 0
    packed = ops.frames_pack_tiles(rgba, cap, bg, tile_hint=hint)
    head = lambda w: w[:64].view(torch.int32)
    assert torch.equal(head(wire)[:8], head(packed)[:8])                      # magic, count, cap, F, T, H, W, bg
    off = lambda w: w[64 + 4 * Fs: 64 + 4 * Fs + 4 * tiles].view(torch.int32)
    assert torch.equal(off(wire) >= 0, off(packed) >= 0)                      # the same tiles are stored
    assert sorted(off(wire)[off(wire) >= 0].tolist()) == list(range(count))   # slots are a permutation of 0 .. count-1
    # too small a capacity: tiles are dropped, the header says so, the unpack raises its flag
    small = count // 2
    wire2 = torch.zeros(ops.frames_wire_bytes(Fs, H, W, small), dtype=torch.uint8, device="cuda")
    raster(full_clip, sel=sel, clamp_output=True, bg=bg, wire=(wire2, small))
    assert ops.frames_wire_count(wire2) == (count, small)
    _, status = ops.frames_unpack_tiles(wire2[None], 1, Fs, H, W, small)
    assert int(status.item()) == 1
    # a frame size with partial edge tiles (the payload's out-of-image pixels read as background, as in the pack pass)
    g = {k: v[sel][:4] for k, v in full_clip["g"].items()}
    Hs, Ws = 200, 300
    view, proj, tanfov, _ = ops.camera_from_intrinsics(full_clip["cam"]["intrinsic"][0, :4].float() * (Ws / W),
                                                       full_clip["cam"]["extrinsic"][0, :4].float(), Hs, Ws)
    t2 = 4 * ((Hs + 15) // 16) * ((Ws + 15) // 16)
    wire3 = torch.zeros(ops.frames_wire_bytes(4, Hs, Ws, t2), dtype=torch.uint8, device="cuda")
    o3 = ops.rasterize(g["xyz"], g["rot"], g["scale"], g["opacity"], g["color"], view, proj, tanfov, Hs, Ws,
                       apply_activations=True, clamp_output=True, wire=(wire3, t2))
    dense3, status = ops.frames_unpack_tiles(wire3[None], 1, 4, Hs, Ws, t2)
    assert torch.equal(dense3, ops.frames_to_rgb8(o3["rgba"])) and int(status.item()) == 0


def test_differential_unpack_equals_the_full_unpack_over_a_sequence_of_steps(full_clip):
    """amav_frames_unpack_tiles_delta into a REUSED dense buffer == amav_frames_unpack_tiles into a fresh one, for a
    sequence in which tiles appear, disappear and the background colour changes; an unchanged step rewrites nothing
    but the stored tiles (checked by poisoning a background tile: it must survive)."""
    from audio_motion_avatar_amd import ops

    nb, Fs = 3, 8
    steps = [(slice(0, 24), (1.0, 1.0, 1.0)), (slice(0, 24), (1.0, 1.0, 1.0)), (slice(100, 124), (1.0, 1.0, 1.0)),
             (slice(30, 54), (0.0, 0.5, 1.0)), (slice(30, 54), (0.0, 0.5, 1.0)), (slice(0, 24), (1.0, 1.0, 1.0))]
    out = torch.empty(nb * Fs, H, W, 3, dtype=torch.uint8, device="cuda").random_(0, 255)  # garbage to start with
    state = ops.frames_tile_state(nb, Fs, H, W, "cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    for i, (sel, bg) in enumerate(steps):
        rgba = raster(full_clip, sel=sel, clamp_output=True, bg=bg)["rgba"].view(nb, Fs, H, W, 4)
        count = max(ops.frames_wire_count(ops.frames_pack_tiles(rgba[b], 0, bg))[0] for b in range(nb))
        wires = torch.stack([ops.frames_pack_tiles(rgba[b], count, bg) for b in range(nb)])
        want, _ = ops.frames_unpack_tiles(wires, nb, Fs, H, W, count)
        if i == 1:  # same frames again: a background tile must not be touched
            assert bool((want[0, :16, :16] == 255).all())
            out[0, :16, :16] = 7
        ops.frames_unpack_tiles(wires, nb, Fs, H, W, count, out=out, status=status, state=state)
        if i == 1:
            assert bool((out[0, :16, :16] == 7).all()), "an unchanged background tile was rewritten"
            out[0, :16, :16] = 255
        assert torch.equal(out, want), f"step {i}"
    assert int(status.item()) == 0
    # a truncated sender: the dropped tiles read as background and the status flag is raised, as in the full unpack
    rgba = raster(full_clip, sel=slice(0, 24), clamp_output=True)["rgba"].view(nb, Fs, H, W, 4)
    count = min(ops.frames_wire_count(ops.frames_pack_tiles(rgba[b], 0))[0] for b in range(nb)) // 2
    wires = torch.stack([ops.frames_pack_tiles(rgba[b], count) for b in range(nb)])
    want, st_full = ops.frames_unpack_tiles(wires, nb, Fs, H, W, count)
    ops.frames_unpack_tiles(wires, nb, Fs, H, W, count, out=out, status=status, state=state)
    assert torch.equal(out, want) and int(status.item()) == 1 and int(st_full.item()) == 1


def test_render_step_is_hip_graph_capturable(full_clip):
    """no entry point of the C ABI allocates or synchronises, so one pass of the hot path (slab
    projection + camera + LBS + fused decode + clear + binning + sort + blend, ONE stream, kernel nodes only) captures
    into a HIP graph; the replay reproduces the eager frames bit for bit, also after the inputs change in place and
    after eager launches of the same entry points between two replays -- the sequence that faulted in round 1 while the
    step still cleared its status words with hipMemsetAsync (memset nodes of a linear graph are replayed from
    captured AQL packets on ROCm 7.2 and an eager hipMemsetAsync in between redirects them: tools/graph_abort_probe.*,
    profiles/r02_graph_abort_probe.txt)."""
    r, tokens, smpl, cam = (full_clip[k] for k in ("renderer", "tokens", "smpl", "cam"))
    Fg = 24
    tok = tokens[0, :Fg].clone()
    sp = {k: v[:, :Fg].clone() for k, v in smpl.items()}
    cm = {k: v[:, :Fg].clone() for k, v in cam.items()}
    ws = [None]
    with torch.no_grad():
        eager, _ = r.render_tokens(tok, sp, cm, workspaces=ws)  # sizes the workspace
        eager = eager.clone()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                out, _ = r.render_tokens(tok, sp, cm, workspaces=ws, check_overflow=False)
        torch.cuda.current_stream().wait_stream(side)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager), "replay differs from the eager frames"
        assert not ws[0].status()[1]
        # new pose in the same buffers -> replay renders the new frames (an eager pass of the same step in between)
        sp["global_orient"].add_(0.3)
        want, _ = r.render_tokens(tok, sp, cm, workspaces=[None])
        want = want.clone()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want) and not torch.equal(want, eager), "replay did not follow the in-place change"
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want)


def _capture(fn):
    """fn() captured into a HIP graph on a side stream (kernel nodes only: the library clears with zero_async)."""
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            out = fn()
    torch.cuda.current_stream().wait_stream(side)
    return graph, out


def test_measured_bounds_attention_is_hip_graph_capturable():
    """VERDICT r2 item 2 / ADVICE: amav_selfattn_forward without proven bounds clears its 16-byte magnitude header
    before the absmax pre-pass.  That clear was a hipMemsetAsync -- the ingredient of round 1's replay fault (memset node
    of a linear graph + an eager memset of the same entry point between two replays).  It is a zero-fill kernel now: the
    measured-bounds call captures, replays bit-identically, follows in-place input changes, with eager calls of the
    same entry point between replays.  Run once per change (a fault here is diagnosed, not retried)."""
    from audio_motion_avatar_amd import ops

    g = torch.Generator().manual_seed(3)
    B, S, H, D = 1, 700, 2, 64
    qkv = torch.randn(B, S, 3 * H * D, generator=g).cuda()
    i = H * D
    call = lambda: ops.selfattn(qkv[..., :i], qkv[..., i:2 * i], qkv[..., 2 * i:], H)  # bounds=None: measured
    with torch.no_grad():
        eager = call().clone()
        torch.cuda.synchronize()
        graph, out = _capture(call)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager)
        qkv.mul_(3.0)                      # new magnitudes: the replayed absmax pass must see them
        want = call().clone()              # eager call of the same entry point between two replays
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want) and not torch.equal(want, eager)
        for _ in range(3):
            call()
            graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want)
    ref = torch.nn.functional.scaled_dot_product_attention(
        *(t.view(B, S, H, D).transpose(1, 2).double() for t in (qkv[..., :i], qkv[..., i:2 * i], qkv[..., 2 * i:])))
    # inputs were scaled by 3 (|v| up to 13): the kernel's 2e-5 bar is relative to the output's magnitude
    assert float((out.view(B, S, H, D).transpose(1, 2).double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_refiner_convolution_is_hip_graph_capturable():
    """Same property for amav_subm_pair_gemm_split (every point-refiner convolution): its scratch clear is a kernel."""
    import numpy as np

    from audio_motion_avatar_amd import ops, point_transformer as pt

    g = torch.Generator().manual_seed(5)
    F_, N, cin, cout = 2, 600, 64, 64
    d = torch.nn.functional.normalize(torch.randn(F_, N, 3, generator=g), dim=-1) * torch.tensor([0.3, 0.5, 0.2])
    n = F_ * N
    cloud_of = torch.arange(F_, dtype=torch.int32).repeat_interleave(N).cuda()
    grid, depth = ops.cloud_voxelize(d.reshape(n, 3).cuda(), cloud_of, F_)
    level = pt.Level(grid, cloud_of, depth, np.full(F_, N), ops.cloud_codes(grid, cloud_of, depth))
    conv = pt.SubMConv3d(cin, cout, 3, bias=True).cuda()
    feat = torch.randn(n, cin, generator=g).cuda()
    with torch.no_grad():
        eager = conv(feat, level).clone()   # builds the pair tables and the split weights (host syncs: outside capture)
        torch.cuda.synchronize()
        graph, out = _capture(lambda: conv(feat, level))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager)
        feat.mul_(40.0)
        want = conv(feat, level).clone()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want) and not torch.equal(want, eager)
        for _ in range(3):
            conv(feat, level)
            graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want)
