"""Camera helpers restated from the reference (pure torch; these functions ARE importable-free restatements).

Follows src/utils/graphic_utils.py:67-78 (getWorld2View2_torch), :103-136 (getProjectionMatrix_torch),
:144-145 (focal2fov_torch) and their use in src/models/renderer.py:486-510 (render_one).
Test infrastructure only (see oracle/__init__.py).
"""
import math

import torch


def focal2fov(focal, pixels):
    """graphic_utils.py:144-145."""
    return 2 * torch.atan(pixels / (2 * focal))


def world2view2(R: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """graphic_utils.py:67-78 with translate=0, scale=1: builds Rt then inverts it twice."""
    Rt = torch.zeros(4, 4, dtype=R.dtype)
    Rt[:3, :3] = R.transpose(0, 1)
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = torch.inverse(Rt)
    cam_center = C2W[:3, 3].clone()
    C2W[:3, 3] = cam_center
    return torch.inverse(C2W)


def projection_matrix(znear, zfar, K: torch.Tensor, w, h) -> torch.Tensor:
    """graphic_utils.py:124-134: K -> NDC projection (the fov arguments are unused by the reference)."""
    fx, fy = K[0, 0].item(), K[1, 1].item()
    px, py = K[0, 2].item(), K[1, 2].item()
    return torch.tensor(
        [
            [2 * fx / w, 0, (2 * px - w) / w, 0],
            [0, 2 * fy / h, (2 * py - h) / h, 0],
            [0, 0, zfar / (zfar - znear), -zfar * znear / (zfar - znear)],
            [0, 0, 1, 0],
        ]
    ).to(K.dtype)


def camera_setup(K: torch.Tensor, E: torch.Tensor, height: int, width: int):
    """renderer.py:486-510 -> (viewmatrix^T [4,4], full_proj^T [4,4], tanfovx, tanfovy, campos [3]).

    The returned matrices are what the reference hands to GaussianRasterizationSettings: row-major tensors that
    the rasterizer reads column-major (renderer.py:507-509).
    """
    R = E[:3, :3].reshape(3, 3).transpose(1, 0)
    T = E[:3, 3]
    znear, zfar = 0.01, 100.0
    FovY = focal2fov(K[1, 1], height)
    FovX = focal2fov(K[0, 0], width)
    tanfovx = math.tan(FovX * 0.5)
    tanfovy = math.tan(FovY * 0.5)
    world_view = world2view2(R, T).transpose(0, 1)
    proj = projection_matrix(znear, zfar, K, width, height).transpose(0, 1)
    full_proj = world_view.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
    campos = world_view.inverse()[3, :3]
    return world_view, full_proj, tanfovx, tanfovy, campos
