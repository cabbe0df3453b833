"""Stage-1 identity encoder on the MI355X: the deterministic segment / z-buffer kernels
(csrc/splat.hip) against the CPU oracle, and the product modules against the vectors produced by running the
reference's own SMPLXTriplaneEncoder / FeatureFusionNetwork (tests/golden/ref_stage1.npz, tier 2)."""
from types import SimpleNamespace

import pytest
import torch

from helpers import ref_fixture, seeded_params, toy_body

pytestmark = pytest.mark.gpu


def close(got, want, rel, what):
    got = got.detach().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = max(1.0, float(want.abs().max()))
    err = float((got.double() - want.double()).abs().max())
    assert err <= rel * scale, f"{what}: max abs {err:.3e} > {rel:.1e} * {scale:.3g}"


def test_cell_reductions_match_the_scatter_restatements_and_are_deterministic():
    from audio_motion_avatar_amd import ops
    from oracle import triplane_net as o_tn

    g = torch.Generator().manual_seed(0)
    B, N, C, R = 2, 700, 48, 6
    feat = torch.randn(B, N, C, generator=g)
    verts = torch.randn(B, N, 3, generator=g) * 0.8
    verts[:, :5] = torch.tensor([[1.4, -1.4, 0.0], [5.0, 0.0, -5.0], [0.0, 0.0, 0.0], [-1.4, 1.4, 1.4], [1.39999, 0.7, -0.7]])
    index = o_tn.cell_indices(verts, 1.4, R)
    cell_of = torch.stack([index[k][:, 0] for k in ("xy", "xz", "yz")], dim=1).to(torch.int32)
    want_pool = o_tn.pool_local(index, feat, R)
    got_pool = ops.cell_pool_max(feat.cuda(), cell_of.cuda(), R * R)
    assert torch.equal(got_pool.cpu(), want_pool)                     # max and a fixed-order 3-term sum: exact
    assert torch.equal(got_pool, ops.cell_pool_max(feat.cuda(), cell_of.cuda(), R * R))
    for p, key in enumerate(("xy", "xz", "yz")):
        want = o_tn.scatter_mean(feat.permute(0, 2, 1), index[key], R * R)
        got = ops.cell_splat_mean(feat.cuda(), cell_of[:, p].contiguous().cuda(), R * R)
        close(got, want, 1e-6, f"scatter_mean {key}")
        assert torch.equal(got, ops.cell_splat_mean(feat.cuda(), cell_of[:, p].contiguous().cuda(), R * R))
        empty = (torch.bincount(index[key][0, 0], minlength=R * R) == 0)
        assert bool((got[0, :, empty.cuda()] == 0).all())


def test_points_projection_kernel_matches_the_oracle_rule():
    from audio_motion_avatar_amd import ops
    from oracle import triplane_net as o_tn

    g = torch.Generator().manual_seed(1)
    B, N, C, H, W = 2, 300, 7, 40, 56
    pts = torch.randn(B, N, 3, generator=g) * torch.tensor([0.5, 0.4, 0.3]) + torch.tensor([0.0, 0.0, 2.0])
    pts[:, :3, 2] = -1.0                                              # behind the camera: never visible
    E = torch.eye(4).repeat(B, 1, 1)
    E[1, :3, 3] = torch.tensor([0.1, -0.05, 0.2])
    K = torch.tensor([[60.0, 0, W / 2], [0, 62.0, H / 2], [0, 0, 1]]).repeat(B, 1, 1)
    feat = torch.randn(B, C, H, W, generator=g)
    want = o_tn.points_projection(pts, E, K, feat, 2.5)
    got = ops.points_project(pts.cuda(), E.cuda(), K.cuda(), feat.cuda(), 2.5).cpu()
    assert torch.equal(got, want)                                     # pure selection: exact
    hit = (want.abs().sum(-1) > 0).float().mean().item()
    assert 0.2 < hit < 0.95 and bool((want[:, :3] == 0).all())


def _product_encoder(meta, body):
    from audio_motion_avatar_amd.smplx_decoder import SMPLXDecoder
    from audio_motion_avatar_amd.triplane_net import FeatureFusionNetwork, SMPLXTriplaneEncoder

    cfg = SimpleNamespace(**meta["cfg"])

    class Encoder(SMPLXTriplaneEncoder):  # the same substitution the fixture's generator made for smplx.SMPLX
        def init_smplx_model(self):
            return body

    enc = Encoder(cfg, SMPLXDecoder(cfg)).eval()
    assert {k: list(v.shape) for k, v in enc.state_dict().items()} == meta["params_encoder"]
    enc.load_state_dict(seeded_params(meta["params_encoder"], "smplx_triplane_encoder."))
    fus = FeatureFusionNetwork(cfg).eval()
    assert {k: list(v.shape) for k, v in fus.state_dict().items()} == meta["params_fusion"]
    fus.load_state_dict(seeded_params(meta["params_fusion"], "fusion_network."))
    return enc.cuda(), fus.cuda()


def test_encoder_and_fusion_network_equal_the_reference_run():
    a, meta, tier = ref_fixture("stage1")
    assert tier == 2
    body = toy_body(**meta["toy_body"]).cuda()
    enc, fus = _product_encoder(meta, body)
    B, T = a["img_tokens"].shape[:2]
    cam = {"intrinsic": torch.zeros(B, T, 3, 3).cuda(), "extrinsic": torch.zeros(B, T, 4, 4).cuda()}
    with torch.no_grad():
        planes, smpl_tokens, pred = enc(cam, a["img_tokens"].cuda(), None, None)
        close(smpl_tokens, a["smpl_tokens"], 2e-5, "smpl predictor tokens")
        for k, v in pred.items():
            close(v, a["pred_" + k], 5e-5, f"predicted {k}")
        # feed the reference's own predictions back as ground truth: the point network alone, on identical vertices
        ref_pred = {k[5:]: v.cuda() for k, v in a.items() if k.startswith("pred_")}
        planes_same, _, _ = enc(cam, a["img_tokens"].cuda(), ref_pred, None)
        close(planes_same, a["planes"], 2e-5, "geometry triplanes")
        planes_gt, _, _ = enc(cam, a["img_tokens"].cuda(), {k: v * 0.5 for k, v in ref_pred.items()}, None)
        close(planes_gt, a["planes_gt"], 2e-5, "geometry triplanes (scaled parameters)")
        fused, smpl_out = fus(a["planes"].cuda(), a["img_tokens"].cuda(), a["smpl_tokens"].cuda())
        close(fused, a["fused"], 2e-5, "fused triplane tokens")
        close(smpl_out, a["smpl_out"], 2e-5, "fused smpl tokens")


def test_triplane_gaussian_avatar_forward_signature():
    """TriplaneGaussianAvatar(cfg).forward(img, smpl_params_gt, cam_params[, image_tokens]) -> the reference's 7-tuple
    (lightning_model_wrapper.py:41-53); without tokens it refuses (the Sapiens encoder is not part of this build)."""
    from audio_motion_avatar_amd._lib import AmavError
    from audio_motion_avatar_amd.config import Stage1Config
    from audio_motion_avatar_amd.synthetic import init_random_heads, make_render_inputs
    from audio_motion_avatar_amd.triplane_net import TriplaneGaussianAvatar

    torch.manual_seed(0)
    H, W = 64, 48
    cfg = Stage1Config(image_size=(H, W), subdivide_steps=0, smplx_transformer_layers=1, cross_transformer_layers=1,
                       device="cuda")
    model = TriplaneGaussianAvatar(cfg).eval()
    init_random_heads(model.renderer)
    with torch.no_grad():
        for blk in model.smplx_triplane_encoder.blocks:
            blk.fc_1.weight.normal_(0, 0.02)   # the reference zero-initialises them
    B, T = 1, 2
    _, smpl, cam = make_render_inputs(T, cfg, seed=4, batch=B)
    img = torch.rand(B, T, 3, H, W, device="cuda")
    tokens = torch.randn(B, T, 4096, 1536, device="cuda") * 0.5
    with pytest.raises(AmavError, match="Sapiens"):
        model(img, smpl, cam)
    with torch.no_grad():
        out = model(img, smpl, cam, image_tokens=tokens)
        again = model(img, smpl, cam, image_tokens=tokens)
    rendered, gaussians, fused, image_tokens, pred1, pred2, smpl_tokens = out
    assert rendered.shape == (B, T, H, W, 3) and fused.shape == (B, T, 256, 3 * 32 * 32) and smpl_tokens.shape == (B, T, 256, 80)
    assert set(gaussians) == {"xyz", "scale", "rot", "opacity", "color", "shs"} and image_tokens.shape == (B, T, 4096, 1536)
    assert pred1["body_pose"].shape == (B, T, 21, 3) and pred2["betas"].shape == (B, T, 10)
    assert torch.isfinite(rendered).all() and torch.isfinite(fused).all()
    assert torch.equal(rendered, again[0]) and torch.equal(fused, again[2])   # deterministic (no atomics-ordered sums)
    # the fused tokens seed stage 2 (main2.py:170-177): they have the layout AudioTriplaneNet consumes
    assert fused[:, -2:].shape == (B, 2, 256, 3072)
