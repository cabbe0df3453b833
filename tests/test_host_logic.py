"""Host-side logic that needs no GPU: head-weight packing, sharding, configuration, synthetic inputs."""
import torch

from audio_motion_avatar_amd import ops
from audio_motion_avatar_amd.config import AudioNetConfig, ModelConfig, RendererConfig
from audio_motion_avatar_amd.dist import shard_range


def test_pack_head_weights_layout():
    C = 4
    g = torch.Generator().manual_seed(0)
    heads = {n: (torch.randn(k, 3 * C + 3, generator=g), torch.randn(k, generator=g))
             for n, k in (("xyz_layer", 3), ("rotation_layer", 4), ("scaling_layer", 3), ("opacity_layer", 1),
                          ("shs_layer", 3))}
    w_plane, w_point = ops.pack_head_weights(heads, C, "cpu")
    assert w_plane.shape == (3, C, 16) and w_point.shape == (16, 4)
    rows = {"xyz_layer": 0, "opacity_layer": 3, "rotation_layer": 4, "scaling_layer": 8, "shs_layer": 12}
    x = torch.randn(3 * C + 3, generator=g)
    out = (w_plane.permute(2, 0, 1).reshape(16, 3 * C) @ x[3:]) + w_point[:, :3] @ x[:3] + w_point[:, 3]
    for n, o in rows.items():
        w, b = heads[n]
        assert torch.allclose(out[o:o + w.shape[0]], w @ x + b, atol=1e-5), n
    assert torch.count_nonzero(out[[11, 15]]) == 0   # pad channels


def test_shard_range_partitions_every_frame_once():
    for total, world in ((2000, 8), (250, 4), (7, 3), (5, 8)):
        got = [shard_range(total, world, r) for r in range(world)]
        assert got[0][0] == 0 and got[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
        assert max(e - s for s, e in got) - min(e - s for s, e in got) <= 1
    assert shard_range(2000, 8, 3) == (750, 1000)   # BASELINE configs[3]: 250 frames per GPU


def test_config_defaults_follow_the_reference_yaml():
    r, a = RendererConfig(), AudioNetConfig()
    assert (r.triplane_resolution, r.triplane_feature_dim, r.radius) == (32, 256, 1.4)
    assert (r.smpl_token_len, r.smpl_token_dim, r.num_expression_coeffs, r.flat_hand_mean) == (80, 256, 10, True)
    assert (a.triplane_input_frames, a.triplane_output_frames) == (2, 6)
    assert (a.transformer_layers, a.transformer_head_dim, a.transformer_num_heads, a.audio_feature_dim) == (8, 64, 8, 768)
    assert ModelConfig().model.triplane_audio_net.smpl_token_len == 80


def test_synthetic_inputs_are_seeded_and_shaped():
    from audio_motion_avatar_amd.synthetic import make_render_inputs

    cfg = RendererConfig(triplane_feature_dim=8, triplane_resolution=4, image_size=(64, 48))
    t1, s1, c1 = make_render_inputs(3, cfg, seed=5, device="cpu")
    t2, s2, c2 = make_render_inputs(3, cfg, seed=5, device="cpu")
    assert t1.shape == (1, 3, 8, 48) and torch.equal(t1, t2) and torch.equal(s1["body_pose"], s2["body_pose"])
    assert s1["body_pose"].shape == (1, 3, 21, 3) and s1["transl"].shape == (1, 3, 3)
    assert c1["intrinsic"].shape == (1, 3, 3, 3) and c1["extrinsic"].shape == (1, 3, 4, 4)
    assert float(c1["intrinsic"][0, 0, 0, 2]) == 24.0 and float(c1["intrinsic"][0, 0, 1, 2]) == 32.0


def _small_model_cfg(**renderer_kw):
    a = AudioNetConfig(triplane_feature_dim=32, triplane_resolution=4, smpl_token_len=6, smpl_token_dim=32,
                       transformer_layers=1, transformer_head_dim=64, transformer_num_heads=1, audio_feature_dim=16,
                       triplane_output_frames=2)
    r = RendererConfig(triplane_feature_dim=32, triplane_resolution=4, smpl_token_len=6, smpl_token_dim=32,
                       image_size=(32, 32), subdivide_steps=0, device="cpu", **renderer_kw)
    return ModelConfig(triplane_audio_net=a, renderer=r)


def test_reference_checkpoint_loads_the_upsampler_and_refuses_silent_gaps():
    """with cfg.upsample_triplane=True (the reference's default renderer.yaml) the
    `triplane_upsampler.*` tensors of a reference checkpoint must be loaded, not dropped; a checkpoint that carries a
    module's prefix but lacks some of its tensors must raise instead of leaving them randomly initialised."""
    import pytest

    from audio_motion_avatar_amd.harness import AudioDrivenAvatar

    torch.manual_seed(0)
    src = AudioDrivenAvatar(_small_model_cfg(upsample_triplane=True, num_upsample_blocks=1))
    with torch.no_grad():
        for p in src.parameters():
            p.add_(torch.randn_like(p) * 0.1)
        for m in src.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.3)
    state = {"audio_triplane." + k: v.clone() for k, v in src.audio_triplane.state_dict().items()}
    state.update({"triplane_gaussian.renderer." + k: v.clone() for k, v in src.renderer.state_dict().items()})
    state["triplane_gaussian.renderer.point_refiner.0.weight"] = torch.zeros(4, 4)      # PTv3 rows: ignored
    state["triplane_gaussian.renderer.smplx_model.v_template"] = torch.zeros(3, 3)      # SMPL-X file buffers: ignored
    state["triplane_gaussian.sapiens_encoder.x"] = torch.zeros(1)                        # stage 1: ignored
    assert any(k.startswith("triplane_gaussian.renderer.triplane_upsampler.") for k in state)

    dst = AudioDrivenAvatar(_small_model_cfg(upsample_triplane=True, num_upsample_blocks=1))
    before = {k: v.clone() for k, v in dst.renderer.triplane_upsampler.state_dict().items()}
    res = dst.load_reference_checkpoint({"state_dict": state})
    assert not res.missing_keys and not res.unexpected_keys
    after = dst.renderer.triplane_upsampler.state_dict()
    assert all(torch.equal(after[k], src.renderer.triplane_upsampler.state_dict()[k]) for k in after)
    assert any(not torch.equal(before[k], after[k]) for k in after)  # the upsampler weights really changed
    for k, v in src.audio_triplane.state_dict().items():
        assert torch.equal(dst.audio_triplane.state_dict()[k], v), k

    # a renderer built WITHOUT the upsampler drops those keys instead of reporting them
    plain = AudioDrivenAvatar(_small_model_cfg())
    res = plain.load_reference_checkpoint({"state_dict": state})
    assert not res.missing_keys and not res.unexpected_keys
    # a checkpoint that has the prefix but lacks tensors of a module that runs here: loud failure
    broken = {k: v for k, v in state.items() if "triplane_upsampler.upsample_blocks.0.upsample.1" not in k}
    with pytest.raises(KeyError, match="randomly initialised"):
        AudioDrivenAvatar(_small_model_cfg(upsample_triplane=True, num_upsample_blocks=1)).load_reference_checkpoint(
            {"state_dict": broken})
    # the audio net only (no renderer entries at all) is accepted unless strict
    only_audio = {k: v for k, v in state.items() if k.startswith("audio_triplane.") and ".renderer." not in k}
    assert not plain.load_reference_checkpoint({"state_dict": only_audio}).missing_keys
    with pytest.raises(KeyError):
        plain.load_reference_checkpoint({"state_dict": only_audio}, strict=True)


def test_smplx_file_with_ten_shape_and_ten_expression_components(tmp_path):
    """smplx reads the expression directions at 10:20 when the model file has fewer than 400 shape
    components (SMPL-X v1.0 SMPLX_NEUTRAL.npz), at 300:300+n otherwise."""
    import numpy as np

    from audio_motion_avatar_amd import body_model as bm

    rng = np.random.default_rng(0)
    V, J = 40, bm.NUM_JOINTS
    base = dict(v_template=rng.normal(size=(V, 3)), f=rng.integers(0, V, size=(60, 3)),
                posedirs=rng.normal(size=(V, 3, (J - 1) * 9)), J_regressor=rng.random((J, V)),
                weights=rng.random((V, J)), kintree_table=np.stack([np.arange(J) - 1, np.arange(J)]).astype(np.int64),
                hands_meanl=np.zeros(45), hands_meanr=np.zeros(45))
    small = dict(base, shapedirs=rng.normal(size=(V, 3, 20)))
    big = dict(base, shapedirs=rng.normal(size=(V, 3, 400)))
    np.savez(tmp_path / "small.npz", **small)
    np.savez(tmp_path / "big.npz", **big)
    a = bm._npz_arrays(str(tmp_path / "small.npz"), 10, 10, True)
    assert a["shapedirs"].shape == (V, 3, 10) and a["expr_dirs"].shape == (V, 3, 10)
    assert np.array_equal(a["expr_dirs"], small["shapedirs"][:, :, 10:20])
    b = bm._npz_arrays(str(tmp_path / "big.npz"), 10, 10, True)
    assert np.array_equal(b["expr_dirs"], big["shapedirs"][:, :, 300:310])
    assert np.array_equal(b["shapedirs"], big["shapedirs"][:, :, :10])


def test_reference_checkpoint_keeps_refiner_keys():
    """point_encoder.* / point_refiner.* are loaded when the Renderer has the refiner (spconv's older
    [k,k,k,C_in,C_out] kernel layout is accepted too) and dropped when it has not."""
    from audio_motion_avatar_amd.harness import AudioDrivenAvatar

    ptv3 = dict(stride=(2,), enc_depths=(1, 1), enc_channels=(32, 64), enc_num_head=(2, 4), enc_patch_size=(64, 64),
                dec_depths=(1,), dec_channels=(32,), dec_num_head=(2,), dec_patch_size=(64,))
    torch.manual_seed(1)
    src = AudioDrivenAvatar(_small_model_cfg(no_point_refiner=False, **ptv3))
    with torch.no_grad():
        for p in src.parameters():
            p.add_(torch.randn_like(p) * 0.1)
    state = {"triplane_gaussian.renderer." + k: v.clone() for k, v in src.renderer.state_dict().items()}
    key = "triplane_gaussian.renderer.point_encoder.point_transformer.embedding.stem.conv.weight"
    want = state[key].clone()
    assert want.shape == (32, 5, 5, 5, 96)
    state[key] = want.permute(1, 2, 3, 4, 0).contiguous()  # spconv 1.x / 2.0 layout
    dst = AudioDrivenAvatar(_small_model_cfg(no_point_refiner=False, **ptv3))
    res = dst.load_reference_checkpoint({"state_dict": state})
    assert not res.missing_keys and not res.unexpected_keys
    got = dst.renderer.state_dict()
    assert torch.equal(got["point_encoder.point_transformer.embedding.stem.conv.weight"], want)
    for k, v in src.renderer.state_dict().items():
        assert torch.equal(got[k], v), k
    assert any(k.startswith("point_refiner.") for k in got) and any(".dec.dec0.up.proj_skip.1.running_var" in k for k in got)
    bare = AudioDrivenAvatar(_small_model_cfg())
    res = bare.load_reference_checkpoint({"state_dict": state})  # refiner keys dropped, nothing missing
    assert not res.missing_keys and not res.unexpected_keys


def test_windowed_upsampler_equals_full_planes_where_it_claims_to():
    """TriplaneUpsampler.forward_tokens_windowed: inside the active tiles plan_windows() reports, the cropped / tiled
    evaluation equals the full one (3 (2^k - 1) texels of halo after k blocks, 2 texels for the last block's tiles),
    incl. tiles at the plane border (last block on the whole crop), and sampling at the points gives the same
    features (the only consumer of the planes)."""
    from types import SimpleNamespace

    from audio_motion_avatar_amd.renderer import TriplaneUpsampler
    from oracle import triplane as o_tri

    for n_blocks, R in ((1, 32), (2, 32), (3, 16)):
        cfg = SimpleNamespace(triplane_feature_dim=8, num_upsample_blocks=n_blocks)
        up = TriplaneUpsampler(cfg).eval()
        for m in up.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.3)
                m.running_var.uniform_(0.5, 1.5)
        assert up.halo_texels() == 3 * (2 ** n_blocks - 1)
        g = torch.Generator().manual_seed(n_blocks)
        tokens = torch.randn(2, 8, 3 * R * R, generator=g)
        radius = 1.4
        tile = 4 * 2 ** n_blocks
        with torch.no_grad():
            full = up.forward_tokens(tokens, R)
            r_out = R * 2 ** n_blocks
            fv = full.view(2, 8, 3, r_out, r_out)
            # a body-like box off centre (tiles), one that touches the -x / +z borders (whole crop), one that needs all
            for case, (lo, hi) in enumerate((((-0.25, -0.6, -0.1), (0.3, 0.55, 0.2)), ((-1.4, -0.2, 0.9), (-1.0, 0.1, 1.4)),
                                             ((-1.3, -1.3, -1.3), (1.3, 1.3, 1.3)))):
                up._window_sizes, up._tile_batch = [[0, 0] for _ in range(3)], {}
                pts = torch.rand(2, 300, 3, generator=g) * (torch.tensor(hi) - torch.tensor(lo)) + torch.tensor(lo)
                plan = up.plan_windows(pts, R, radius)
                assert up.windows_contain(plan, pts, R, radius)
                if case == 0:
                    assert all(w["tiles"] is not None and 0 < len(w["tiles"]) < 2 * (R // 4) ** 2 for w in plan)
                if case == 1:
                    assert all(w["tiles"] is not None for w in plan)
                    assert any(int(w["tiles"][:, 1:].min()) == 0 or int(w["tiles"][:, 1:].max()) == R // 4 - 1 for w in plan)  # border tiles
                got = up.forward_tokens_windowed(tokens, R, plan, out=torch.full_like(full, float("nan")))
                gv = got.view(2, 8, 3, r_out, r_out)
                for p, w in enumerate(plan):
                    for f, ty, tx in torch.nonzero(w["mask"]).tolist():
                        a = gv[f, :, p, ty * tile:(ty + 1) * tile, tx * tile:(tx + 1) * tile]
                        b = fv[f, :, p, ty * tile:(ty + 1) * tile, tx * tile:(tx + 1) * tile]
                        assert torch.isfinite(a).all() and (a - b).abs().max() <= 1e-5, (n_blocks, case, p, f, ty, tx)
                planes_w = o_tri.tokens_to_planes(torch.nan_to_num(got)[None], r_out)
                planes_f = o_tri.tokens_to_planes(full[None], r_out)
                fw = o_tri.sample_from_triplane(planes_w, pts, radius)
                ff = o_tri.sample_from_triplane(planes_f, pts, radius)
                assert (fw - ff).abs().max() <= 1e-5
                # a point outside the planned tiles is reported
                far = pts.clone()
                far[0, 0] = torch.tensor(hi) + torch.tensor([0.5, 0.5, -0.5])
                if case == 0:
                    assert not up.windows_contain(plan, far, R, radius)
            # a margin activates the neighbouring tiles too; crop sizes only grow
            up._window_sizes = [[0, 0] for _ in range(3)]
            one = torch.zeros(1, 1, 3) + 0.01  # (per-frame masks: a frame's tiles are its own)
            two = up.plan_windows(torch.stack([one[0], one[0] + 0.6]), R, radius)
            assert all(not torch.equal(w["mask"][0], w["mask"][1]) for w in two)
            up._window_sizes = [[0, 0] for _ in range(3)]
            tight = up.plan_windows(one, R, radius)
            wide = up.plan_windows(one, R, radius, margin=1.0)
            assert all(int(w["mask"].sum()) >= int(t["mask"].sum()) for w, t in zip(wide, tight))
            assert any(int(w["mask"].sum()) > int(t["mask"].sum()) for w, t in zip(wide, tight))
            again = up.plan_windows(one, R, radius)
            size = lambda pl: [(w["crop"][1] - w["crop"][0], w["crop"][3] - w["crop"][2]) for w in pl]
            assert size(again) == size(wide)


def test_windowed_upsampler_degenerate_plans():
    """Resolution not a multiple of the tile size, non-finite points, and a plane nobody samples: whole planes / nothing,
    never a wrong tile."""
    from types import SimpleNamespace

    from audio_motion_avatar_amd.renderer import TriplaneUpsampler

    up = TriplaneUpsampler(SimpleNamespace(triplane_feature_dim=4, num_upsample_blocks=2)).eval()
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        R = 6  # not a multiple of TILE_CELLS: every plane as a whole
        tokens = torch.randn(1, 4, 3 * R * R, generator=g)
        pts = torch.rand(1, 50, 3, generator=g) - 0.5
        plan = up.plan_windows(pts, R, 1.4)
        assert all(w["tiles"] is None and w["crop"] == (0, R, 0, R) for w in plan)
        assert torch.allclose(up.forward_tokens_windowed(tokens, R, plan, out=torch.zeros(1, 4, 3 * 24 * 24)),
                              up.forward_tokens(tokens, R), atol=1e-6)
        assert up.windows_contain(plan, pts * 100, R, 1.4)  # whole planes contain everything
        R = 8
        tokens = torch.randn(1, 4, 3 * R * R, generator=g)
        bad = pts.clone()
        bad[0, 3, 1] = float("nan")
        plan = up.plan_windows(bad, R, 1.4)
        assert all(w["tiles"] is None for w in plan)
        # points far outside the radius clamp onto the border texels: those tiles are planned (and exact)
        far = torch.full((1, 5, 3), 50.0)
        plan = up.plan_windows(far, R, 1.4)
        got = up.forward_tokens_windowed(tokens, R, plan, out=torch.full((1, 4, 3 * 32 * 32), float("nan")))
        full = up.forward_tokens(tokens, R)
        gv, fv = got.view(1, 4, 3, 32, 32), full.view(1, 4, 3, 32, 32)
        for p, w in enumerate(plan):
            assert w["mask"][0, -1, -1] and int(w["mask"].sum()) == 1
            assert (gv[0, :, p, -16:, -16:] - fv[0, :, p, -16:, -16:]).abs().max() <= 1e-5


def test_weight_caches_work_under_inference_mode():
    """tensors created under torch.inference_mode() (the fused q/k/v weight on a first forward inside it)
    track no version counter; the memo keys must not read `_version` on them."""
    from audio_motion_avatar_amd.transformer import Attention, _memo

    attn = Attention(query_dim=32, heads=2, dim_head=16)  # parameters are ordinary tensors, as in a loaded model
    with torch.inference_mode():
        w = attn._qkv_weight()
        assert w.is_inference() and w.shape == (96, 32)
        assert attn._qkv_weight() is w                      # cached, no exception
        calls = []
        make = lambda: calls.append(1) or w.sum()
        a = _memo("probe", (w,), make)
        b = _memo("probe", (w,), make)
        assert a is b and len(calls) == 1
    # parameters built outside inference mode still invalidate the cache when modified in place
    attn2 = Attention(query_dim=32, heads=2, dim_head=16)
    w1 = attn2._qkv_weight()
    with torch.no_grad():
        attn2.to_q.weight.add_(1.0)
    assert attn2._qkv_weight() is not w1
    assert ops.tensor_version(w) == -1 and ops.tensor_version(attn2.to_q.weight) >= 1


def test_differential_unpack_limits():
    """the delta unpack keeps a per-frame tile table in LDS; frames with more tiles (or a width that is not a
    multiple of 16) must take the full unpack instead of failing on every step."""
    assert ops.frames_delta_unpack_supported(512, 512)
    assert ops.frames_delta_unpack_supported(1296, 2304)          # TED frames: 81 x 144 = 11 664 tiles
    assert not ops.frames_delta_unpack_supported(1440, 2560)      # 90 x 160 = 14 400 > 14 336
    assert not ops.frames_delta_unpack_supported(2160, 3840)      # 32 400 tiles
    assert not ops.frames_delta_unpack_supported(512, 520)        # width % 16 != 0
    assert ops.DELTA_UNPACK_MAX_TILES == 14336


def test_renderer_refuses_a_training_mode_upsampler():
    """crops, tile mosaics and batch chunks change what a training-mode BatchNorm2d would average over; the
    renderer is inference-only and says so instead of silently computing other statistics."""
    import pytest

    from audio_motion_avatar_amd._lib import AmavError
    from audio_motion_avatar_amd.renderer import Renderer
    from audio_motion_avatar_amd.synthetic import make_render_inputs

    cfg = RendererConfig(image_size=(32, 32), subdivide_steps=0, triplane_feature_dim=8, triplane_resolution=8,
                         predict_smplx_params=False, upsample_triplane=True, num_upsample_blocks=1, device="cpu")
    r = Renderer(cfg)          # nn.Module default: training mode
    tokens, smpl, cam = make_render_inputs(2, cfg, seed=1, device="cpu")
    with pytest.raises(AmavError, match="training mode"):
        r(tokens, cam, torch.zeros(1, 2, 1, 1), smpl)
    before = r.triplane_upsampler.upsample_blocks[0].upsample[3].block[0].running_mean.clone()
    assert torch.equal(before, r.triplane_upsampler.upsample_blocks[0].upsample[3].block[0].running_mean)
