"""Pins the triplane / decoder / transformer oracles: analytic properties, torch's own modules, golden fixtures."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import smplx_decoder as o_dec, transformer as o_tr, triplane as o_tri

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_constant_plane_samples_constant_inside_and_half_at_the_border():
    """align_corners=False + zero padding: interior taps sum to 1, a point exactly on the +-1 border sees half."""
    C, R = 4, 8
    planes = torch.ones(1, 3, C, R, R)
    pts = torch.tensor([[[0.1, -0.2, 0.3], [1.4, 0.0, 0.0], [5.0, 5.0, 5.0]]])  # radius 1.4 -> normalised 1.0 clamps
    f = o_tri.sample_from_triplane(planes, pts, 1.4)
    assert torch.allclose(f[0, 0], torch.ones(3 * C))
    # x = +1: planes 0 (x,y) and 1 (x,z) straddle the border in x (half weight); plane 2 (y,z) is interior
    assert torch.allclose(f[0, 1], torch.cat([torch.full((2 * C,), 0.5), torch.ones(C)]))
    assert torch.allclose(f[0, 2], torch.full((3 * C,), 0.25))  # clamped corner: quarter weight on every plane


def test_plane_axes_follow_the_reference():
    """plane 0 <- (x, y), plane 1 <- (x, z), plane 2 <- (y, z); grid x indexes W (renderer.py:300-310)."""
    R = 16
    planes = torch.zeros(1, 3, 1, R, R)
    w = torch.arange(R).float()
    planes[0, 0, 0] = w[None, :].expand(R, R)   # value = column index (driven by x)
    planes[0, 1, 0] = w[:, None].expand(R, R)   # value = row index (driven by z)
    planes[0, 2, 0] = w[None, :].expand(R, R)   # value = column index (driven by y)
    p = torch.tensor([[[0.35, -0.7, 0.14]]])
    f = o_tri.sample_from_triplane(planes, p, 1.4)[0, 0]
    pix = lambda g: ((g + 1) * R - 1) / 2
    assert torch.allclose(f, torch.tensor([pix(0.25), pix(0.1), pix(-0.5)]), atol=1e-5)


def test_triplane_golden_fixture():
    g = np.load(os.path.join(GOLD, "triplane_r8c16.npz"))
    t = lambda k: torch.from_numpy(g[k])
    params = {k[2:]: t(k) for k in g.files if k.startswith("p_")}
    planes = o_tri.tokens_to_planes(t("tokens")[None], 8)
    out = o_tri.decode_gaussians(params, planes, t("points"), t("transl"), float(g["radius"]))
    for k in ("xyz", "scale", "rot", "opacity", "color"):
        assert (out[k] - t("out_" + k)).abs().max() < 1e-5, k
    assert (o_tri.sample_from_triplane(planes, t("points"), float(g["radius"])) - t("features")).abs().max() < 1e-5
    assert torch.allclose(out["rot"].norm(dim=-1), torch.ones(2, 256), atol=1e-6)


def test_single_key_cross_attention_ignores_the_queries():
    """The audio context has length 1 (triplane_audio_net.py:211): attn2 is a broadcast of to_out(to_v(audio))."""
    g = torch.Generator().manual_seed(0)
    dim, ctx = 64, 48
    p = {"a.to_q.weight": torch.randn(dim, dim, generator=g), "a.to_k.weight": torch.randn(dim, ctx, generator=g),
         "a.to_v.weight": torch.randn(dim, ctx, generator=g), "a.to_out.0.weight": torch.randn(dim, dim, generator=g),
         "a.to_out.0.bias": torch.randn(dim, generator=g)}
    enc = torch.randn(2, 1, ctx, generator=g)
    y1 = o_tr.attention(p, "a.", torch.randn(2, 10, dim, generator=g), enc, heads=2)
    y2 = o_tr.attention(p, "a.", torch.randn(2, 10, dim, generator=g), enc, heads=2)
    expect = torch.nn.functional.linear(torch.nn.functional.linear(enc, p["a.to_v.weight"]), p["a.to_out.0.weight"],
                                        p["a.to_out.0.bias"])
    assert torch.allclose(y1, y2, atol=1e-5) and torch.allclose(y1, expect.expand(2, 10, dim), atol=1e-5)


def test_transformer_block_golden_fixture():
    g = np.load(os.path.join(GOLD, "transformer_block_small.npz"))
    t = lambda k: torch.from_numpy(g[k])
    params = {k[2:]: t(k) for k in g.files if k.startswith("p_")}
    y = o_tr.transformer_block(params, "b.", t("x"), t("enc"), heads=2)
    assert (y - t("y")).abs().max() < 2e-4


def test_temporal_reducers_match_the_torch_modules_they_restate():
    torch.manual_seed(1)
    C, T = 16, 2
    conv = nn.Conv3d(3 * C, 3 * C, (T, 1, 1), groups=3 * C, bias=False)
    x = torch.randn(2, T, 3, C, 4, 4)
    ref = conv(x.permute(0, 2, 3, 1, 4, 5).contiguous().view(2, 3 * C, T, 4, 4)).view(2, 3, C, 1, 4, 4)
    got = o_tr.triplane_temporal_reducer({"m.conv_time.weight": conv.weight}, "m.", x)
    assert torch.allclose(got, ref.permute(0, 3, 1, 2, 4, 5), atol=1e-6)
    # closed form: a per-channel two-tap weighted sum over time
    w = conv.weight.view(3, C, T)
    assert torch.allclose(got[:, 0], torch.einsum("btpchw,pct->bpchw", x, w), atol=1e-6)

    D = 32
    mha = nn.MultiheadAttention(D, 8, dropout=0.1, batch_first=True).eval()
    mlp = nn.Sequential(nn.Linear(D, 2 * D), nn.ReLU(), nn.Linear(2 * D, D))
    n1, n2 = nn.LayerNorm(D), nn.LayerNorm(D)
    xs = torch.randn(2, T, D, 5)
    import einops
    h = einops.rearrange(xs, "b t c s -> (b s) t c")
    a, _ = mha(h, h, h)
    h = n1(h + a)
    h = n2(h + mlp(h)).mean(dim=1, keepdim=True)
    ref = einops.rearrange(h, "(b s) t c -> b t c s", b=2)
    p = {"s.self_attn.in_proj_weight": mha.in_proj_weight, "s.self_attn.in_proj_bias": mha.in_proj_bias,
         "s.self_attn.out_proj.weight": mha.out_proj.weight, "s.self_attn.out_proj.bias": mha.out_proj.bias,
         "s.mlp.0.weight": mlp[0].weight, "s.mlp.0.bias": mlp[0].bias, "s.mlp.2.weight": mlp[2].weight,
         "s.mlp.2.bias": mlp[2].bias, "s.norm1.weight": n1.weight, "s.norm1.bias": n1.bias,
         "s.norm2.weight": n2.weight, "s.norm2.bias": n2.bias}
    with torch.no_grad():
        assert torch.allclose(o_tr.smplx_temporal_reducer(p, "s.", xs), ref, atol=1e-5)


def test_smplx_decoder_shapes_and_rotation_validity():
    g = torch.Generator().manual_seed(2)
    D, L = 8, 5
    shapes = {"mlp.0": (1024, D * L), "mlp.2": (512, 1024), "mlp.4": (256, 512), "dec_body_root_pose": (6, 256),
              "dec_body_pose": (126, 256), "dec_body_shape": (10, 256), "dec_transl": (3, 256),
              "dec_hand_pose": (180, 256), "dec_face_expression": (10, 256), "dec_face_jaw_pose": (6, 256),
              "dec_leye_pose": (6, 256), "dec_reye_pose": (6, 256)}
    p = {}
    for k, (o, i) in shapes.items():
        p[f"smpl_decoder.{k}.weight"] = torch.randn(o, i, generator=g) / i ** 0.5
        p[f"smpl_decoder.{k}.bias"] = torch.randn(o, generator=g) * 0.1
    out = o_dec.smplx_decoder_forward(p, torch.randn(3, D, L, generator=g))
    want = {"betas": (3, 10), "transl": (3, 3), "global_orient": (3, 3), "body_pose": (3, 21, 3),
            "left_hand_pose": (3, 15, 3), "right_hand_pose": (3, 15, 3), "jaw_pose": (3, 3), "leye_pose": (3, 3),
            "reye_pose": (3, 3), "expression": (3, 10)}
    assert {k: tuple(v.shape) for k, v in out.items()} == want
    assert all(torch.isfinite(v).all() for v in out.values())
    assert out["body_pose"].norm(dim=-1).max() <= np.pi + 1e-4   # axis-angle of a proper rotation


def test_product_smplx_decoder_matches_oracle_on_cpu():
    """The product module is plain torch (library GEMMs), so it can be checked against the oracle without a GPU."""
    from audio_motion_avatar_amd.config import RendererConfig
    from audio_motion_avatar_amd.smplx_decoder import SMPLXDecoder

    torch.manual_seed(3)
    cfg = RendererConfig(smpl_token_dim=8, smpl_token_len=5)
    dec = SMPLXDecoder(cfg).eval()
    tokens = torch.randn(4, 8, 5)
    with torch.no_grad():
        got = dec(tokens)
        ref = o_dec.smplx_decoder_forward({"smpl_decoder." + k: v for k, v in dec.state_dict().items()}, tokens)
    assert set(got) == set(ref)
    for k in ref:
        assert got[k].shape == ref[k].shape, k
        assert torch.allclose(got[k], ref[k], atol=1e-5), k
