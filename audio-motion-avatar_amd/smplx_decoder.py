"""SMPLXDecoder: 80x256 SMPL-X tokens -> SMPL-X parameters (mirror of src/models/smplx_decoder.py:8-145).

Same module / parameter names as the reference (`mlp.{0,2,4}`, `dec_*`), so `smpl_decoder.*` checkpoint keys load.
The three Linear+ReLU layers and nine small heads are plain library GEMMs (rocBLAS through torch); the rot6d ->
axis-angle conversions replace pytorch3d.transforms (absent) with batched torch ops on the device, all 55 joints
of all frames in one pass instead of the reference's nine separate conversions.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def rotation_6d_to_matrix(d6: torch.Tensor) -> torch.Tensor:
    """Gram-Schmidt rows (pytorch3d convention; smplx_decoder.py:106-123)."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = F.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=-1)), dim=-2)


def matrix_to_axis_angle(m: torch.Tensor) -> torch.Tensor:
    """pytorch3d's default path: matrix -> quaternion (largest-denominator candidate, w >= 0) -> axis-angle."""
    lead = m.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.reshape(lead + (9,)).unbind(-1)
    q_abs = torch.sqrt(torch.clamp_min(torch.stack(
        [1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], -1), 0.0))
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], -1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], -1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], -1)], -2)
    cand = cand / (2.0 * q_abs[..., None].clamp_min(0.1))
    pick = q_abs.argmax(-1)
    quat = torch.gather(cand, -2, pick[..., None, None].expand(lead + (1, 4))).squeeze(-2)
    quat = torch.where(quat[..., :1] < 0, -quat, quat)
    norms = torch.norm(quat[..., 1:], p=2, dim=-1, keepdim=True)
    half = torch.atan2(norms, quat[..., :1])
    angles = 2 * half
    small = angles.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(angles), angles)
    sin_half_over_angle = torch.where(small, 0.5 - angles * angles / 48, torch.sin(half) / safe)
    return quat[..., 1:] / sin_half_over_angle


class SMPLXDecoder(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        self.cfg = cfg
        self.smpl_token_dim = cfg.smpl_token_dim
        self.smpl_token_len = cfg.smpl_token_len
        self.mlp = nn.Sequential(
            nn.Linear(self.smpl_token_dim * self.smpl_token_len, 1024), nn.ReLU(),
            nn.Linear(1024, 512), nn.ReLU(),
            nn.Linear(512, 256), nn.ReLU())
        self.body_joint_num = 22
        self.hand_joint_num = 15
        self.shape_dim = 10
        self.expression_dim = cfg.num_expression_coeffs
        self.dec_body_root_pose = nn.Linear(256, 6)
        self.dec_body_pose = nn.Linear(256, (self.body_joint_num - 1) * 6)
        self.dec_body_shape = nn.Linear(256, self.shape_dim)
        self.dec_transl = nn.Linear(256, 3)
        self.dec_hand_pose = nn.Linear(256, 2 * self.hand_joint_num * 6)
        self.dec_face_expression = nn.Linear(256, self.expression_dim)
        self.dec_face_jaw_pose = nn.Linear(256, 6)
        self.dec_leye_pose = nn.Linear(256, 6)
        self.dec_reye_pose = nn.Linear(256, 6)

    def forward(self, tokens):
        B = tokens.shape[0]
        feat = self.mlp(tokens.reshape(B, -1))
        h = self.hand_joint_num
        # all 55 rotations in one conversion: root(1) body(21) lhand(15) rhand(15) jaw leye reye
        d6 = torch.cat([self.dec_body_root_pose(feat), self.dec_body_pose(feat), self.dec_hand_pose(feat),
                        self.dec_face_jaw_pose(feat), self.dec_leye_pose(feat), self.dec_reye_pose(feat)], dim=1)
        aa = matrix_to_axis_angle(rotation_6d_to_matrix(d6.reshape(B, 55, 6)))
        return {
            "betas": self.dec_body_shape(feat),
            "transl": self.dec_transl(feat),
            "global_orient": aa[:, 0],
            "body_pose": aa[:, 1:22],
            "left_hand_pose": aa[:, 22:22 + h],
            "right_hand_pose": aa[:, 22 + h:22 + 2 * h],
            "jaw_pose": aa[:, 52],
            "leye_pose": aa[:, 53],
            "reye_pose": aa[:, 54],
            "expression": self.dec_face_expression(feat),
        }
