"""Image and SMPL-X parameter metrics of the reference (src/utils/loss_utils.py), as the demo prints them per window
(src/main2.py:205-211) and the training step sums them (src/models/lightning_model_wrapper.py:495-530).

Same names, arguments and reductions as the reference, evaluated where the tensors live (on the MI355X for rendered
frames: elementwise work and an 11 x 11 depthwise window through the library's convolution -- none of it is on the
rendering hot path, so there is no hand-written kernel here).  Pinned by reference-run fixtures: l1 / l2 / ssim / window
exactly as shipped (tier 1), the geodesic and SMPL-X parameter losses with smplx's absent `batch_rodrigues` restated
(tier 2) -- tests/test_reference_golden.py.  `LPIPS` needs the `lpips` package and its VGG weights, neither of which is
available offline: constructing it raises.
"""
import math

import torch
import torch.nn.functional as F


def l1_loss(network_output, gt):
    """loss_utils.py:18-19"""
    return torch.abs(network_output - gt).mean()


def l2_loss(network_output, gt):
    """loss_utils.py:21-22"""
    return ((network_output - gt) ** 2).mean()


def psnr(network_output, gt):
    """10 log10(1 / MSE) over everything (BASELINE.json's quality metric; the reference implements no PSNR)."""
    return 10.0 * torch.log10(1.0 / l2_loss(network_output, gt).clamp_min(1e-20))


def gaussian(window_size, sigma):
    """loss_utils.py:24-26: normalised 1-D Gaussian taps."""
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return g / g.sum()


def create_window(window_size, channel):
    """loss_utils.py:28-32: [channel, 1, window, window] separable Gaussian (sigma 1.5)."""
    w1 = gaussian(window_size, 1.5).unsqueeze(1)
    w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous()


def _ssim(img1, img2, window, window_size, channel, size_average=True):
    """loss_utils.py:63-84"""
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean() if size_average else ssim_map.mean(1).mean(1).mean(1)


def ssim(img1, img2, window_size=11, size_average=True):
    """loss_utils.py:44-61: img1, img2 [B, T, H, W, C] (the renderer's frame layout)."""
    img1 = img1.reshape(-1, *img1.shape[2:]).permute(0, 3, 1, 2)
    img2 = img2.reshape(-1, *img2.shape[2:]).permute(0, 3, 1, 2)
    channel = img1.size(1)
    window = create_window(window_size, channel).to(img1.device).type_as(img1)
    return _ssim(img1, img2, window, window_size, channel, size_average)


class LPIPS(torch.nn.Module):
    """loss_utils.py:87-105 wraps lpips.LPIPS(net='vgg'): a pretrained network that cannot be fetched here."""

    def __init__(self):
        super().__init__()
        raise RuntimeError("LPIPS needs the `lpips` package and its pretrained VGG weights, which are not available offline; "
                           "l1_loss / ssim / psnr are implemented")


def batch_rodrigues(rot_vecs, epsilon=1e-8):
    """smplx.lbs.batch_rodrigues (absent here): axis-angle [N,3] -> [N,3,3]."""
    angle = torch.norm(rot_vecs + epsilon, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos, sin = torch.cos(angle)[:, None], torch.sin(angle)[:, None]
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros_like(rx)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view(-1, 3, 3)
    ident = torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device)[None]
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def rotation_geodesic_loss(rot_vec_pred, rot_vec_gt):
    """loss_utils.py:109-135: mean geodesic angle between two sets of axis-angle rotations [..., 3]."""
    if rot_vec_pred.shape != rot_vec_gt.shape:
        raise AssertionError(f"Shape mismatch: {rot_vec_pred.shape} vs {rot_vec_gt.shape}")
    if rot_vec_pred.shape[-1] != 3:
        raise AssertionError("the last dimension must be 3")
    R_pred = batch_rodrigues(rot_vec_pred.reshape(-1, 3))
    R_gt = batch_rodrigues(rot_vec_gt.reshape(-1, 3))
    RT = torch.matmul(R_pred.transpose(1, 2), R_gt)
    cos = (torch.diagonal(RT, dim1=1, dim2=2).sum(-1) - 1) / 2
    return torch.acos(torch.clamp(cos, -0.999, 0.999)).mean()


ROTATION_KEYS = ("global_orient", "body_pose", "left_hand_pose", "right_hand_pose", "jaw_pose", "leye_pose", "reye_pose")


def smplx_param_loss(pred_params, gt_params, weights=None):
    """loss_utils.py:137-182 -> (total, dict of the parts under the reference's names)."""
    if weights is None:
        weights = {k: 1.0 for k in ("betas",) + ROTATION_KEYS + ("expression", "transl")}
    losses = {}
    total = 0.0
    if "betas" in pred_params and "betas" in gt_params:
        losses["betas_mse"] = F.mse_loss(pred_params["betas"], gt_params["betas"])
        losses["betas_prior"] = torch.mean(pred_params["betas"] ** 2)
        total = total + weights["betas"] * losses["betas_mse"] + 0.01 * losses["betas_prior"]
    for key in ROTATION_KEYS:
        if key in pred_params and key in gt_params:
            losses[f"{key}_geo"] = rotation_geodesic_loss(pred_params[key], gt_params[key])
            total = total + weights.get(key, 1.0) * losses[f"{key}_geo"]
    if "expression" in pred_params and "expression" in gt_params:
        losses["expression_l1"] = F.l1_loss(pred_params["expression"], gt_params["expression"])
        losses["expression_prior"] = torch.mean(pred_params["expression"] ** 2)
        total = total + weights["expression"] * losses["expression_l1"] + 0.01 * losses["expression_prior"]
    if "transl" in pred_params and "transl" in gt_params:
        losses["transl_smoothl1"] = F.smooth_l1_loss(pred_params["transl"], gt_params["transl"])
        total = total + weights["transl"] * losses["transl_smoothl1"]
    return total, losses
