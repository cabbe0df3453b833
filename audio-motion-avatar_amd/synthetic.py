"""Seeded synthetic inputs of the BASELINE.json configs (there is no dataset, checkpoint or SMPL-X file here).

Shapes and distributions follow SURVEY.md section 8(d): triplane tokens ~ N(0,1) in the reference token layout
[B,T,C,3R^2] (renderer.py:85-91), head weights ~ N(0, 0.02^2) on top of the reference bias initialisation
(renderer.py:57-71), per-frame pose jitter ~ N(0, 0.2^2) around an upright body, TED-style camera (E = I,
dataset_speech_vid.py:306-317) looking at a body 2.4 m away with fx = fy = W.
"""
import math

import torch


def init_random_heads(renderer, seed=42, std=0.02):
    """Random head weights (the reference's all-zero init would make every Gaussian identical)."""
    g = torch.Generator().manual_seed(seed)
    gd = renderer.gaussian_decoder
    with torch.no_grad():
        for layer in (gd.xyz_layer, gd.rotation_layer, gd.scaling_layer, gd.opacity_layer, gd.shs_layer):
            layer.weight.copy_(torch.randn(layer.weight.shape, generator=g) * std)
        gd.xyz_layer.weight.mul_(0.05)  # keep the offsets at the millimetre scale of a trained model
    return renderer


def make_render_inputs(num_frames, cfg, seed=42, device="cuda", pose_jitter=0.2, batch=1):
    """-> (triplane_tokens [B,T,C,3R^2], smpl_params dict of [B,T,...], cam_params dict) on `device`."""
    g = torch.Generator().manual_seed(seed)
    B, T = batch, num_frames
    C, R = cfg.triplane_feature_dim, cfg.triplane_resolution
    H, W = cfg.image_size
    rn = lambda *s: torch.randn(*s, generator=g)
    tokens = rn(B, T, C, 3 * R * R)
    go = rn(B, T, 3) * (pose_jitter * 0.25)
    go[..., 0] += math.pi  # SMPL-X is y-up, the TED camera is y-down: turn the body upright in the image
    hands = pose_jitter * 0.5
    smpl = {
        "global_orient": go,
        "body_pose": rn(B, T, 21, 3) * pose_jitter,
        "betas": rn(B, T, 10) * 0.5,
        "left_hand_pose": rn(B, T, 15, 3) * hands,
        "right_hand_pose": rn(B, T, 15, 3) * hands,
        "jaw_pose": rn(B, T, 3) * 0.05,
        "leye_pose": rn(B, T, 3) * 0.02,
        "reye_pose": rn(B, T, 3) * 0.02,
        "expression": rn(B, T, cfg.num_expression_coeffs) * 0.5,
        "transl": torch.tensor([0.0, -0.15, 2.4]).expand(B, T, 3) + rn(B, T, 3) * 0.02,
    }
    K = torch.tensor([[float(W), 0.0, W / 2.0], [0.0, float(W), H / 2.0], [0.0, 0.0, 1.0]]).expand(B, T, 3, 3)
    E = torch.eye(4).expand(B, T, 4, 4)
    to = lambda t: t.contiguous().to(device)
    return to(tokens), {k: to(v) for k, v in smpl.items()}, {"intrinsic": to(K), "extrinsic": to(E)}
