"""Triplane sampling, Gaussian heads and Renderer.forward, restated on CPU.

Follows src/models/renderer.py:73-204 (Renderer.forward), :292-317 (sample_from_triplane, which calls
torch's own F.grid_sample), :165-171 (five linear heads) and :319-346 (construct_gaussians).
Parameters are passed as a state-dict-like mapping with the reference's names
(`gaussian_decoder.xyz_layer.weight`, ...; renderer.py:51-55).  Test infrastructure only (oracle/__init__.py).
"""
import einops
import torch
import torch.nn.functional as F

HEADS = ("xyz_layer", "rotation_layer", "scaling_layer", "opacity_layer", "shs_layer")


def tokens_to_planes(triplane_tokens, resolution):
    """renderer.py:85-91: [B,T,C,(Np Hp Wp)] -> [(B T),3,C,R,R]."""
    return einops.rearrange(triplane_tokens, "B T Ct (Np Hp Wp) -> (B T) Np Ct Hp Wp", Np=3, Hp=resolution,
                            Wp=resolution)


def sample_from_triplane(triplane_features, points, radius):
    """renderer.py:292-317.  triplane_features [B,3,C,R,R], points [B,N,3] -> [B,N,3C]."""
    batched = points.ndim == 3
    if not batched:
        triplane_features = triplane_features[None, ...]
        points = points[None, ...]
    positions = torch.clamp(points / radius, -1, 1)
    indices2D = torch.stack((positions[..., [0, 1]], positions[..., [0, 2]], positions[..., [1, 2]]), dim=-3)
    out = F.grid_sample(
        einops.rearrange(triplane_features, "B Np Cp Hp Wp -> (B Np) Cp Hp Wp", Np=3),
        einops.rearrange(indices2D, "B Np N Nd -> (B Np) () N Nd", Np=3),
        align_corners=False,
        mode="bilinear",
    )
    out = einops.rearrange(out, "(B Np) Cp () N -> B N (Np Cp)", Np=3)
    if not batched:
        out = out.squeeze(0)
    return out


def gaussian_heads(params, decoder_input, prefix="gaussian_decoder."):
    """renderer.py:167-171."""
    lin = lambda name: F.linear(decoder_input, params[prefix + name + ".weight"], params[prefix + name + ".bias"])
    return dict(xyz_offset=lin("xyz_layer"), rotation=lin("rotation_layer"), scaling=lin("scaling_layer"),
                opacity=lin("opacity_layer"), shs=lin("shs_layer"))


def construct_gaussians(gaussian_params, points, transl):
    """renderer.py:319-346.  transl [F,3]."""
    rotation = F.normalize(gaussian_params["rotation"], dim=-1)
    color = torch.sigmoid(gaussian_params["shs"])
    return dict(xyz=points + gaussian_params["xyz_offset"] + transl.reshape(-1, 1, 3),
                scale=gaussian_params["scaling"], rot=rotation, opacity=gaussian_params["opacity"], color=color,
                shs=color)


def refine_points(params, triplane_features, points, radius, ptv3_cfg):
    """renderer.py:136-151: initial features -> PTv3Encoder -> point_refiner MLP -> points + offsets (oracle/ptv3.py)."""
    from . import ptv3

    feats = sample_from_triplane(triplane_features, points, radius)
    point_features = ptv3.encoder_forward(params, "point_encoder.", points, feats, ptv3_cfg)
    return points + ptv3.point_refiner(params, "point_refiner.", point_features).reshape(points.shape)


def decode_gaussians(params, triplane_features, points, transl, radius, ptv3_cfg=None):
    """renderer.py:136-181.  Without `ptv3_cfg` the point refiner is bypassed (offset == 0: an untrained refiner's
    last layer is zero-initialised, renderer.py:46-47)."""
    if ptv3_cfg is not None:
        points = refine_points(params, triplane_features, points, radius, ptv3_cfg)
    feats = sample_from_triplane(triplane_features, points, radius)
    decoder_input = torch.cat([points, feats], dim=-1)
    return construct_gaussians(gaussian_heads(params, decoder_input), points, transl)


def triplane_upsampler(params, triplanes, num_blocks, prefix="triplane_upsampler.", eps=1e-5):
    """renderer.py:377-417 restated with torch functionals (eval-mode BatchNorm: running statistics).
    triplanes [B,3,C,H,W] -> [B,3,C,2^n H,2^n W]."""
    B, P, C, H, W = triplanes.shape
    cur = triplanes.reshape(B * P, C, H, W)
    skip = cur
    conv = lambda x, name, pad: F.conv2d(x, params[prefix + name + ".weight"], params[prefix + name + ".bias"],
                                         padding=pad)
    bn = lambda x, name: F.batch_norm(x, params[prefix + name + ".running_mean"], params[prefix + name + ".running_var"],
                                      params[prefix + name + ".weight"], params[prefix + name + ".bias"], False, 0.0, eps)
    up = lambda x: F.interpolate(x, scale_factor=2, mode="nearest")
    for i in range(num_blocks):
        b = f"upsample_blocks.{i}.upsample."
        x = F.relu(conv(up(cur), b + "1", 1))
        r = conv(F.relu(bn(x, b + "3.block.0")), b + "3.block.2", 1)
        r = conv(F.relu(bn(r, b + "3.block.3")), b + "3.block.5", 1)
        x = x + r  # ResBlock with identity skip (in == out channels)
        if i == 0:
            skip = conv(skip, "skip_connections.0.0", 0)
        skip = up(skip)
        cur = x + skip
    return cur.reshape(B, P, C, cur.shape[-2], cur.shape[-1])
