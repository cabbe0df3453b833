"""Seeded synthetic inputs shared by the CPU (oracle) and GPU (parity) tests."""
import numpy as np
import torch


def random_scene(seed, N, H, W, F=1, spread=0.35, depth=2.5, log_scale=-3.6, scale_jitter=0.5, focal=None):
    """Activated Gaussians (as GaussianRasterizer receives them) + per-frame K/E."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    xyz = rn(F, N, 3) * spread + torch.tensor([0.0, 0.0, depth])
    rot = torch.nn.functional.normalize(rn(F, N, 4), dim=-1)
    scale = torch.exp(rn(F, N, 3) * scale_jitter + log_scale)
    opacity = torch.sigmoid(rn(F, N, 1) * 1.5)
    color = torch.rand(F, N, 3, generator=g)
    focal = focal or float(W)
    K = torch.tensor([[focal, 0, W / 2], [0, focal, H / 2], [0, 0, 1.0]]).repeat(F, 1, 1)
    K[:, 0, 2] += rn(F) * 2
    K[:, 1, 2] += rn(F) * 2
    E = torch.eye(4).repeat(F, 1, 1)
    ang = rn(F) * 0.1
    E[:, 0, 0], E[:, 0, 2], E[:, 2, 0], E[:, 2, 2] = torch.cos(ang), torch.sin(ang), -torch.sin(ang), torch.cos(ang)
    E[:, :3, 3] = rn(F, 3) * 0.05
    return dict(xyz=xyz, rot=rot, scale=scale, opacity=opacity, color=color, K=K, E=E, H=H, W=W)


def oracle_frames(scene, dtype=np.float32, bg=(1, 1, 1), **settings):
    """Run every frame of `scene` through the C oracle (already-activated inputs)."""
    from oracle import camera, rasterizer

    outs = []
    tdt = torch.float64 if np.dtype(dtype) == np.float64 else torch.float32
    for f in range(scene["xyz"].shape[0]):
        view, proj, tx, ty, _ = camera.camera_setup(scene["K"][f].to(tdt), scene["E"][f].to(tdt), scene["H"],
                                                    scene["W"])
        outs.append(rasterizer.rasterize_c(scene["xyz"][f], scene["rot"][f], scene["scale"][f], scene["opacity"][f],
                                           scene["color"][f], view, proj, tx, ty, bg, scene["H"], scene["W"],
                                           dtype=dtype, **settings))
    return outs


def random_pose(seed, F, scale=0.2, num_joints=55, num_coeffs=20):
    g = torch.Generator().manual_seed(seed)
    pose = torch.randn(F, num_joints * 3, generator=g) * scale
    coeffs = torch.randn(F, num_coeffs, generator=g)
    return pose, coeffs
