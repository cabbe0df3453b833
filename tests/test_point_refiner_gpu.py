"""Point refiner (PTv3) on the GPU against oracle/ptv3.py and the reference-run fixtures ref_ptv3_codes / ref_ptv3.

Integer work (grid, keys, orders, neighbour tables, clusters) is compared bit for bit; floating point within the
tolerance written at each assert (fp32 sums evaluated in a different order than the CPU's)."""
import numpy as np
import pytest
import torch

from helpers import ref_fixture, seeded_params

pytestmark = pytest.mark.gpu


def _mods():
    from audio_motion_avatar_amd import ops, point_transformer

    return ops, point_transformer


def _clouds(seed, F, N, extent=(0.3, 0.5, 0.2)):
    g = torch.Generator().manual_seed(seed)
    d = torch.nn.functional.normalize(torch.randn(F, N, 3, generator=g), dim=-1)
    pts = d * torch.tensor(extent) * (1.0 + 0.05 * torch.randn(F, N, 1, generator=g))
    pts[:, N - N // 8:] = pts[:, : N // 8] + 0.002 * torch.randn(F, N // 8, 3, generator=g)  # shared voxels
    return pts + torch.randn(F, 1, 3, generator=g) * 0.3


def test_codes_bit_exact_against_reference_encode():
    """amav_cloud_codes == the reference's own encode() (tier-1 fixture) for all four orders, depths 1..16."""
    ops, _ = _mods()
    a, meta, _ = ref_fixture("ptv3_codes")
    for depth in meta["depths"]:
        grid = a[f"grid_{depth}"].cuda()
        n = grid.shape[0]
        keys = ops.cloud_codes(grid, torch.zeros(n, dtype=torch.int32, device="cuda"),
                               torch.tensor([depth], dtype=torch.int32, device="cuda")).cpu()
        for k, order in enumerate(meta["orders"]):
            want = a[f"code_{depth}_{order}"] & ((1 << (3 * depth)) - 1)  # the fixture carries batch << 3 depth on top
            assert torch.equal(keys[k], want), (depth, order)


def test_voxelize_keys_neighbors_bit_exact():
    from oracle import ptv3 as o_pt

    ops, _ = _mods()
    F, N = 3, 700
    pts = _clouds(5, F, N)
    cloud_of = torch.arange(F, dtype=torch.int32).repeat_interleave(N)
    grid, depth = ops.cloud_voxelize(pts.reshape(-1, 3).cuda(), cloud_of.cuda(), F)
    keys = ops.cloud_codes(grid, cloud_of.cuda(), depth)
    sorted_keys, order = torch.sort(keys, dim=1, stable=True)
    starts = (torch.arange(F + 1) * N).to(torch.int32).cuda()
    for f in range(F):
        g = o_pt.frame_grid(pts[f])
        assert torch.equal(grid[f * N:(f + 1) * N].cpu().long(), g)
        batch = torch.zeros(N, dtype=torch.long)
        code, o_order, _, d = o_pt.serialization(g, batch)
        assert int(depth[f]) == d
        assert torch.equal(keys[:, f * N:(f + 1) * N].cpu() & ((1 << 48) - 1), code)
        assert torch.equal(order[:, f * N:(f + 1) * N].cpu() - f * N, o_order)
        for ksize in (3, 5):
            nbr = ops.cloud_neighbors(grid, cloud_of.cuda(), depth, starts, sorted_keys[0], order[0], ksize).cpu().long()
            want = o_pt.neighbor_table(g, batch, ksize)
            got = nbr[f * N:(f + 1) * N]
            got = torch.where(got >= 0, got - f * N, got)
            assert torch.equal(got, want), (f, ksize)


@pytest.mark.parametrize("cin,cout,ksize,F,N", [(12, 32, 5, 1, 400), (64, 64, 3, 2, 700), (768, 32, 5, 1, 900),
                                                 (256, 256, 3, 2, 500), (96, 160, 3, 1, 300)])
def test_subm_conv_matches_oracle(cin, cout, ksize, F, N):
    """Gather-GEMM over voxel pairs + ordered sum == the per-tap restatement (fp64), incl. C_in not a multiple of 32,
    C_out tiles of 32 / 64 / 128 and taps whose pair count is not a multiple of the 128-pair tile."""
    from oracle import ptv3 as o_pt

    ops, pt = _mods()
    pts = _clouds(7, F, N)
    n = F * N
    cloud_of = torch.arange(F, dtype=torch.int32).repeat_interleave(N).cuda()
    grid, depth = ops.cloud_voxelize(pts.reshape(n, 3).cuda(), cloud_of, F)
    level = pt.Level(grid, cloud_of, depth, np.full(F, N), ops.cloud_codes(grid, cloud_of, depth))
    gen = torch.Generator().manual_seed(8)
    feat = torch.randn(n, cin, generator=gen)
    conv = pt.SubMConv3d(cin, cout, ksize, bias=True)
    with torch.no_grad():
        conv.bias.copy_(torch.randn(cout, generator=gen) * 0.1)
    got = conv.cuda()(feat.cuda(), level).cpu()
    conv = conv.cpu()
    for f in range(F):
        nbr = o_pt.neighbor_table(o_pt.frame_grid(pts[f]), torch.zeros(N, dtype=torch.long), ksize)
        want = o_pt.subm_conv3d(feat[f * N:(f + 1) * N].double(), nbr, conv.weight.detach().double(), conv.bias.detach().double())
        err = float((got[f * N:(f + 1) * N].double() - want).abs().max())
        assert err <= 1e-5 * max(1.0, float(want.abs().max())), (f, err)
    assert level.pairs(ksize).count == int((level.neighbors(ksize) >= 0).sum())


@pytest.mark.parametrize("scale", [1.0, 1e-4, 3e3])
def test_subm_conv_split_products_match_the_fp32_mfma_form(monkeypatch, scale):
    """Default: three fp16 partial products per fp32 product (pair_gemm_f16_kernel), features scaled by the power of two
    of the call's largest magnitude; AMAV_SUBM=f32: the fp32 MFMA kernel.  Same result to fp32 rounding at any scale."""
    ops, pt = _mods()
    F, N, cin, cout, ksize = 2, 800, 128, 128, 3
    pts = _clouds(11, F, N)
    n = F * N
    cloud_of = torch.arange(F, dtype=torch.int32).repeat_interleave(N).cuda()
    grid, depth = ops.cloud_voxelize(pts.reshape(n, 3).cuda(), cloud_of, F)
    level = pt.Level(grid, cloud_of, depth, np.full(F, N), ops.cloud_codes(grid, cloud_of, depth))
    gen = torch.Generator().manual_seed(9)
    feat = (torch.randn(n, cin, generator=gen) * scale).cuda()
    feat[5] *= 50.0  # one outlier row sets the call's scale; the other rows keep their precision
    conv = pt.SubMConv3d(cin, cout, ksize, bias=True).cuda()
    split = conv(feat, level)
    assert torch.equal(conv(feat, level), split)  # reproducible
    monkeypatch.setenv("AMAV_SUBM", "f32")
    plain = conv(feat, level)
    ref = float(plain.abs().max())
    assert float((split - plain).abs().max()) <= 4e-6 * ref
    with torch.no_grad():
        conv.weight.mul_(0.5)  # the prepared weights follow in-place updates
    monkeypatch.delenv("AMAV_SUBM")
    bias = conv.bias.detach()
    assert float((conv(feat, level) - bias - 0.5 * (split - bias)).abs().max()) <= 4e-6 * ref


@pytest.mark.parametrize("heads,dim,counts,patch", [(4, 64, [1300, 512, 70], 512), (2, 32, [300, 130], 128),
                                                    (2, 16, [257, 33, 64], 128), (1, 64, [5], 512)])
def test_patch_attention_matches_oracle(heads, dim, counts, patch):
    """Padded-patch attention incl. a borrowed tail, clouds smaller than the patch and smaller than an MFMA tile."""
    from oracle import ptv3 as o_pt

    ops, pt = _mods()
    C = heads * dim
    n = sum(counts)
    gen = torch.Generator().manual_seed(11)
    qkv = torch.randn(n, 3 * C, generator=gen)
    order = torch.cat([torch.randperm(c, generator=gen) + s for c, s in zip(counts, np.cumsum([0] + counts[:-1]))])
    level = pt.Level.__new__(pt.Level)
    level.counts, level.starts_host = counts, np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    level.grid, level._patches = qkv.cuda(), {}
    desc, max_patch = level.patches(patch)
    got = ops.patch_attention(qkv.cuda(), order.cuda(), desc, heads, max_patch).cpu()
    want = torch.empty(n, C, dtype=torch.float64)
    start = 0
    for c in counts:
        o = order[start:start + c] - start
        inv = torch.empty_like(o)
        inv[o] = torch.arange(c)
        K, pad, unpad = o_pt.patch_layout(c, patch)
        x = qkv[start:start + c].double()[o[pad]]
        q, k, v = x.reshape(-1, K, 3, heads, dim).permute(2, 0, 3, 1, 4).unbind(0)
        att = torch.softmax((q * dim ** -0.5) @ k.transpose(-2, -1), -1)
        want[start:start + c] = (att @ v).transpose(1, 2).reshape(-1, C)[unpad[inv]]
        start += c
    assert (got.double() - want).abs().max() <= 2e-6 * max(1.0, float(want.abs().max()))


def _reference_net(meta):
    _, pt = _mods()
    cfg = meta["cfg"]
    net = pt.PointTransformerV3(order=pt.ORDERS, **cfg).eval()
    assert {k: list(v.shape) for k, v in net.state_dict().items()} == meta["params"]  # names + shapes of the reference
    net.load_state_dict(seeded_params(meta["params"], "point_encoder.point_transformer."))
    return net.cuda()


def test_network_matches_reference_run():
    """PointTransformerV3 on the HIP path == the reference's classes run on CPU (tier-2 fixture), cloud by cloud."""
    a, meta, _ = ref_fixture("ptv3")
    net = _reference_net(meta)
    for ci in range(meta["clouds"]):
        out = net(a[f"pts_{ci}"][None].cuda(), a[f"feat_{ci}"][None].cuda()).cpu()
        ref = a[f"out_{ci}"]
        err = float((out - ref).abs().max())
        assert err <= 1e-4 * max(1.0, float(ref.abs().max())), (ci, err)


def test_degenerate_clouds_match_oracle():
    """Edge cases of the serialisation: a cloud inside ONE voxel (depth 0: no pooling shift, every tap empty), a
    two-voxel cloud (depth 1), and a ragged pair of tiny clouds smaller than one MFMA tile."""
    from oracle import ptv3 as o_pt

    a, meta, _ = ref_fixture("ptv3")
    net = _reference_net(meta)
    p = {"pe.point_transformer." + k: v for k, v in seeded_params(meta["params"], "point_encoder.point_transformer.").items()}
    g = torch.Generator().manual_seed(31)
    N = 24
    one_voxel = torch.rand(N, 3, generator=g) * 0.009 + 0.5          # floor(100 p) identical for all points
    two_voxels = one_voxel.clone()
    two_voxels[N // 2:, 1] += 0.01
    spread = torch.randn(N, 3, generator=g) * 0.05
    pts = torch.stack([one_voxel, two_voxels, spread])
    feat = torch.randn(3, N, meta["cfg"]["in_channels"], generator=g)
    assert [int(o_pt.frame_grid(c).max()).bit_length() for c in pts][:2] == [0, 1]
    got = net(pts.cuda(), feat.cuda()).cpu()
    want = o_pt.encoder_forward(p, "pe.", pts, feat, meta["cfg"])
    assert torch.isfinite(got).all()
    assert (got - want).abs().max() <= 1e-4 * max(1.0, float(want.abs().max()))


def test_batched_clouds_equal_single_clouds():
    """Clouds of one pass do not see each other: a batch of clouds == each cloud alone (definition 2), and vs oracle."""
    from oracle import ptv3 as o_pt

    a, meta, _ = ref_fixture("ptv3")
    net = _reference_net(meta)
    F, N = 3, 420
    pts = _clouds(21, F, N, extent=(0.2, 0.35, 0.15))
    feat = torch.randn(F, N, meta["cfg"]["in_channels"], generator=torch.Generator().manual_seed(22))
    both = net(pts.cuda(), feat.cuda()).cpu().reshape(F, N, -1)
    for f in range(F):
        alone = net(pts[f:f + 1].cuda(), feat[f:f + 1].cuda()).cpu()
        # not bit-equal: the GEMM library may split a [3N, C] and an [N, C] product differently
        assert (both[f] - alone[0:N]).abs().max() <= 2e-5 * max(1.0, float(alone.abs().max())), f
    p = {"pe.point_transformer." + k: v for k, v in seeded_params(meta["params"], "point_encoder.point_transformer.").items()}
    want = o_pt.encoder_forward(p, "pe.", pts, feat, meta["cfg"]).reshape(F, N, -1)
    assert (both - want).abs().max() <= 1e-4 * max(1.0, float(want.abs().max()))


def test_renderer_with_point_refiner_matches_oracle():
    """cfg.no_point_refiner=False: LBS -> sample -> PTv3 -> MLP -> refined points -> fused decode (renderer.py:127-181)
    against the CPU chain, with a non-zero last refiner layer (the reference zero-initialises it)."""
    from audio_motion_avatar_amd.config import RendererConfig
    from audio_motion_avatar_amd.renderer import Renderer
    from audio_motion_avatar_amd.synthetic import init_random_heads, make_render_inputs
    from oracle import lbs as o_lbs, subdivide as o_sub, triplane as o_tri

    torch.manual_seed(0)
    pcfg = dict(stride=(2, 2), enc_depths=(1, 1, 1), enc_channels=(32, 64, 128), enc_num_head=(2, 4, 4),
                enc_patch_size=(256, 256, 256), dec_depths=(1, 1), dec_channels=(64, 64), dec_num_head=(1, 2),
                dec_patch_size=(256, 256))
    cfg = RendererConfig(image_size=(64, 64), subdivide_steps=0, triplane_feature_dim=16, triplane_resolution=8,
                         predict_smplx_params=False, no_point_refiner=False, num_gaussians=1500, refiner_clouds_per_pass=2,
                         **pcfg)
    r = init_random_heads(Renderer(cfg).eval(), std=0.05)
    assert float(r.point_refiner[-1].weight.detach().abs().max()) == 0.0  # renderer.py:46-47
    with torch.no_grad():
        r.point_refiner[-1].weight.normal_(0, 0.02)
        for m in r.point_encoder.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
    F_ = 3
    tokens, smpl, cam = make_render_inputs(F_, cfg, seed=4)
    with torch.no_grad():
        images, gaussians = r(tokens, cam, torch.zeros(1, F_, 1, 1, device="cuda"), smpl)
        plain = Renderer(RendererConfig(**{**cfg.__dict__, "no_point_refiner": True})).eval()
        plain.gaussian_decoder.load_state_dict(r.gaussian_decoder.state_dict())
        _, g_plain = plain(tokens, cam, torch.zeros(1, F_, 1, 1, device="cuda"), smpl)
    params = {k: v.detach().cpu() for k, v in r.state_dict().items()}
    levels = o_sub.subdivision_levels(r.smplx_model.faces, r.smplx_model.num_verts, 1)
    sp = {k: v.cpu() for k, v in smpl.items()}
    pts = o_lbs.get_smpl_vertices(r.smplx_model.oracle_arrays(torch.float32), sp, densify=(levels, r.subset_index))
    planes = o_tri.tokens_to_planes(tokens.cpu(), cfg.triplane_resolution)
    g = o_tri.decode_gaussians(params, planes, pts, sp["transl"].reshape(-1, 3), cfg.radius,
                               ptv3_cfg={k: list(v) for k, v in pcfg.items()})
    moved = float((gaussians["xyz"] - g_plain["xyz"]).abs().max())
    assert moved > 1e-3, moved  # the refiner did move the points
    for k in ("xyz", "scale", "rot", "opacity", "color"):
        assert (gaussians[k].cpu() - g[k]).abs().max() <= 1e-4, k
    assert images.shape == (1, F_, 64, 64, 3)



def test_windowed_upsampler_with_refiner_and_fallback():
    """upsample_triplane + point refiner: the windowed upsampler (planes cropped to what the points can sample) gives
    the Gaussians of the full-plane evaluation; when the refiner moves points past the planned margin the renderer
    notices and falls back to full planes."""
    from audio_motion_avatar_amd.config import RendererConfig
    from audio_motion_avatar_amd.renderer import Renderer
    from audio_motion_avatar_amd.synthetic import init_random_heads, make_render_inputs

    pcfg = dict(stride=(2,), enc_depths=(1, 1), enc_channels=(32, 64), enc_num_head=(2, 4), enc_patch_size=(256, 256),
                dec_depths=(1,), dec_channels=(32,), dec_num_head=(2,), dec_patch_size=(256,))
    base = dict(image_size=(64, 64), subdivide_steps=0, triplane_feature_dim=16, triplane_resolution=32,
                predict_smplx_params=False, no_point_refiner=False, num_gaussians=1200, upsample_triplane=True,
                num_upsample_blocks=2, radius=2.8, **pcfg)   # radius 2.8: the body fills a quarter of the planes
    r_win = init_random_heads(Renderer(RendererConfig(upsample_windows=True, **base)).eval(), std=0.05)
    r_full = Renderer(RendererConfig(upsample_windows=False, **base)).eval()
    r_full.load_state_dict(r_win.state_dict())
    F_ = 2
    tokens, smpl, cam = make_render_inputs(F_, r_win.cfg, seed=9)
    dummy = torch.zeros(1, F_, 1, 1, device="cuda")
    for shift, expect_fallback in ((0.02, False), (1.5, True)):  # metres; the planned margin is 0.1 m + cell rounding
        with torch.no_grad():
            for r in (r_win, r_full):
                r.point_refiner[-1].weight.normal_(0, 1, generator=None).mul_(0)
                r.point_refiner[-1].bias.fill_(shift)                 # a uniform shift of every point
            r_full.load_state_dict(r_win.state_dict())
            calls = []
            orig = r_win.triplane_upsampler.forward_tokens
            r_win.triplane_upsampler.forward_tokens = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
            _, g_win = r_win(tokens, cam, dummy, smpl)
            r_win.triplane_upsampler.forward_tokens = orig
            _, g_full = r_full(tokens, cam, dummy, smpl)
        assert bool(calls) == expect_fallback, (shift, calls)
        if not expect_fallback:
            plan = r_win.last_window_plan
            assert all(w["tiles"] is not None and 0 < len(w["tiles"]) <= 32 * F_ for w in plan)  # really tiled: <= half
        for k in ("xyz", "scale", "rot", "opacity", "color"):
            assert (g_win[k] - g_full[k]).abs().max() <= 2e-5, (shift, k)
