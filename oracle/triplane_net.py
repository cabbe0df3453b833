"""Stage-1 identity encoder restated on CPU (functional, weights by reference name).  Test infrastructure only.

Follows src/models/triplane_net.py:16-58 (ResnetBlockFC), :124-207 (SMPLXTriplaneEncoder.forward), :226-244
(pool_local / generate_plane_features with torch_scatter's scatter_max / scatter_mean: absent here -> PARITY UNPINNED
for the two scatter ops, restated with torch.scatter_reduce / index_add_), :376-409 (FeatureFusionNetwork.forward),
src/models/tokenizers.py (TriplaneLearnablePositionalEmbedding), src/models/image_feature.py:257-275 (ImageFeature) and
src/utils/graphic_utils.py:275-331 (points_projection on pytorch3d's PointsRasterizer: absent -> PARITY UNPINNED; the
restatement below is the deterministic form the build defines: nearest point per pixel inside a disc of `radius_px`
around the projected point, pixel centres at +0.5, and a point takes the features of the LAST pixel in (y, x) order
where it is the nearest one, which is what a sequential index_put does).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import transformer as o_tr


def resnet_block_fc(p, prefix, x):
    net = F.linear(F.relu(x), p[prefix + "fc_0.weight"], p[prefix + "fc_0.bias"])
    dx = F.linear(F.relu(net), p[prefix + "fc_1.weight"], p[prefix + "fc_1.bias"])
    xs = F.linear(x, p[prefix + "shortcut.weight"]) if prefix + "shortcut.weight" in p else x
    return xs + dx


def scatter_max(src, index, dim_size):
    """torch_scatter.scatter_max(src [B,C,N], index [B,1,N], dim_size)[0]: per-cell maximum, 0 for an empty cell."""
    B, C, N = src.shape
    out = torch.zeros(B, C, dim_size, dtype=src.dtype)
    return out.scatter_reduce(2, index.expand(-1, C, -1), src, reduce="amax", include_self=False)


def scatter_mean(src, index, dim_size):
    """torch_scatter.scatter_mean into zeros: per-cell mean, 0 for an empty cell (sequential sum in point order)."""
    B, C, N = src.shape
    idx = index.expand(-1, C, -1)
    total = torch.zeros(B, C, dim_size, dtype=src.dtype).scatter_add(2, idx, src)
    count = torch.zeros(B, C, dim_size, dtype=src.dtype).scatter_add(2, idx, torch.ones_like(src))
    return total / count.clamp_min(1)


def cell_indices(verts, radius, R):
    """triplane_net.py:163-183 -> dict of int64 [BT,1,N]."""
    pos = (torch.clamp(verts, -radius + 1e-6, radius - 1e-6) + radius) / (2 * radius)
    out = {}
    for key, (a, b) in (("xy", (0, 1)), ("xz", (0, 2)), ("yz", (1, 2))):
        x = (pos[..., [a, b]] * R).long()
        out[key] = torch.clamp(x[..., 0] + R * x[..., 1], 0, R * R - 1)[:, None, :]
    return out


def pool_local(index, c, R):
    """triplane_net.py:226-238."""
    c_out = 0
    for key in ("xy", "xz", "yz"):
        fea = scatter_max(c.permute(0, 2, 1), index[key], R * R)
        c_out = c_out + fea.gather(2, index[key].expand(-1, c.shape[2], -1))
    return c_out.permute(0, 2, 1)


def points_projection(points, w2c, intrinsics, features, radius_px):
    """points [B,N,3], w2c [B,4,4], intrinsics [B,3,3], features [B,C,H,W] -> [B,N,C] (see the module docstring)."""
    B, N, _ = points.shape
    _, C, H, W = features.shape
    out = torch.zeros(B, N, C, dtype=features.dtype)
    pts, E, K = points.double().numpy(), w2c.double().numpy(), intrinsics.double().numpy()
    f32 = np.float32
    for b in range(B):
        zbuf = np.full((H, W), np.inf, dtype=np.float32)
        ids = np.full((H, W), -1, dtype=np.int64)
        Ef, Kf, Pf = E[b].astype(f32), K[b].astype(f32), pts[b].astype(f32)
        for n in range(N):
            p = Pf[n]
            X = f32(f32(f32(Ef[0, 0] * p[0]) + f32(Ef[0, 1] * p[1])) + f32(Ef[0, 2] * p[2])) + Ef[0, 3]
            Y = f32(f32(f32(Ef[1, 0] * p[0]) + f32(Ef[1, 1] * p[1])) + f32(Ef[1, 2] * p[2])) + Ef[1, 3]
            Z = f32(f32(f32(Ef[2, 0] * p[0]) + f32(Ef[2, 1] * p[1])) + f32(Ef[2, 2] * p[2])) + Ef[2, 3]
            if not Z > 0:
                continue
            u, v = f32(f32(Kf[0, 0] * X) / Z) + Kf[0, 2], f32(f32(Kf[1, 1] * Y) / Z) + Kf[1, 2]
            r, half = f32(radius_px), f32(0.5)
            x0, x1 = max(0, int(np.ceil(f32(f32(u - r) - half)))), min(W - 1, int(np.floor(f32(f32(u + r) - half))))
            y0, y1 = max(0, int(np.ceil(f32(f32(v - r) - half)))), min(H - 1, int(np.floor(f32(f32(v + r) - half))))
            for y in range(y0, y1 + 1):
                for x in range(x0, x1 + 1):
                    dx, dy = f32(f32(f32(x) + half) - u), f32(f32(f32(y) + half) - v)
                    inside = f32(f32(dx * dx) + f32(dy * dy)) < f32(r * r)
                    if inside and (Z < zbuf[y, x] or (Z == zbuf[y, x] and n < ids[y, x])):
                        zbuf[y, x], ids[y, x] = Z, n
        for y in range(H):  # sequential index_put: the last pixel a point wins keeps its features
            for x in range(W):
                if ids[y, x] >= 0:
                    out[b, ids[y, x]] = features[b, :, y, x]
    return out


def image_feature(p, prefix, rgb, tokens):
    """image_feature.py:262-275."""
    B, Nv, Nt, C = tokens.shape
    H, W = rgb.shape[-2:]
    side = int(round(Nt ** 0.5))
    f = F.linear(tokens.reshape(-1, C), p[prefix + "feature_reducer.weight"], p[prefix + "feature_reducer.bias"])
    f = f.reshape(B * Nv, side, side, -1).permute(0, 3, 1, 2).contiguous()
    f = F.interpolate(f, size=(H, W), mode="bilinear", align_corners=False)
    return torch.cat([rgb.reshape(B * Nv, *rgb.shape[2:]), f], dim=1).reshape(B, Nv, -1, H, W)


def encoder_forward(p, prefix, verts, verts_feat, radius, R):
    """The point network of SMPLXTriplaneEncoder.forward (:159-198) from posed vertices [BT,N,3] and per-vertex features
    [BT,N,C] to planes [BT,3,C,R,R]."""
    lin = lambda n, x: F.linear(x, p[prefix + n + ".weight"], p[prefix + n + ".bias"])
    net = resnet_block_fc(p, prefix + "blocks.0.", lin("fc_pos", torch.cat([verts, verts_feat], dim=-1)))
    index = cell_indices(verts, radius, R)
    for i in (1, 2):
        net = resnet_block_fc(p, f"{prefix}blocks.{i}.", torch.cat([net, pool_local(index, net, R)], dim=2))
    c = lin("fc_c", net)
    planes = [scatter_mean(c.permute(0, 2, 1), index[k], R * R).reshape(c.shape[0], -1, R, R) for k in ("xy", "xz", "yz")]
    return torch.stack(planes, dim=1)


def fusion_forward(p, prefix, geometry_triplane, image_tokens, smpl_tokens, num_layers, heads):
    """FeatureFusionNetwork.forward (:376-409); attention through oracle/transformer.py (general cross-attention)."""
    B, T, _, C, H, W = geometry_triplane.shape
    geo = geometry_triplane.reshape(B * T, 3, C, H, W) + p[prefix + "triplane_tokenizer_geometry.embeddings"][None]
    geo_tokens = geo.permute(0, 2, 1, 3, 4).reshape(B * T, C, -1)
    combined = torch.cat([geo_tokens, smpl_tokens], dim=2)
    out = o_tr.transformer1d(p, prefix + "transformer_cross.", combined, image_tokens.reshape(B * T, *image_tokens.shape[2:]),
                             num_layers, heads)
    tokens, smpl_out = torch.split(out, [geo_tokens.shape[2], smpl_tokens.shape[2]], dim=2)
    return tokens.reshape(B, T, *tokens.shape[1:]), smpl_out.reshape(B, T, *smpl_out.shape[1:])
