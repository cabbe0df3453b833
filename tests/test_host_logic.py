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
