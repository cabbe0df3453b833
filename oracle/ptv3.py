"""Point refiner (Pointcept PointTransformerV3 as configured by the reference) restated on CPU.  Test infrastructure.

Follows src/models/point_transformer/point_encoder.py:6-40 (PTv3Encoder), pointtransformer_v3.py:81-145
(Point.serialization), :328-499 (SerializedAttention, enable_flash=False), :528-615 (Block), :618-721
(SerializedPooling), :724-759 (SerializedUnpooling), :762-792 (Embedding), :795-991 (PointTransformerV3),
serialization/default.py:10-27, z_order.py:42-118, hilbert.py:93-190 and src/models/renderer.py:34-47,143-151
(point_refiner MLP, refined points).

PINNED BY: tests/golden/ref_ptv3_codes.npz (tier 1: the reference's own `encode` for the four orders) and
tests/golden/ref_ptv3.npz (tier 2: the reference's PointTransformerV3 classes run on CPU with addict.Dict,
torch_scatter.segment_csr and spconv's SubMConv3d / SparseConvTensor injected).  PARITY UNPINNED for `subm_conv3d`
itself: spconv (unlisted, imported at pointtransformer_v3.py:15) is absent and has no ROCm build; the restatement
follows spconv 2.x's published semantics (weight [C_out, k, k, k, C_in]; output sites = input sites; taps centred,
`padding` ignored for submanifold convolutions) and is cross-checked against torch's dense conv3d in
tests/test_oracle_ptv3.py.

DETERMINISTIC SEMANTICS DEFINED BY THE BUILD (the reference defines no reproducible output for this stage: the four
serialisation orders are permuted with an unseeded torch.randperm at every level, pointtransformer_v3.py:137-141,
685-689; `grid_coord = floor(100 p)` is negative for half the body, point_encoder.py:33, which the z-order / Hilbert
encoders and spconv's index hash do not support; the 1 cm voxels hold several points each, for which spconv's hash
insertion keeps an arbitrary one):
  1. the orders keep their configured sequence (z, z-trans, hilbert, hilbert-trans): the reference's own
     shuffle_orders=False path;
  2. every frame is processed as the reference processes a batch of ONE cloud (patch size, serialisation depth and
     grid origin are per frame, so a frame's result does not depend on which frames share its batch -- frames are the
     unit of data parallelism);
  3. grid_coord = floor(100 p) - min over the frame's points of floor(100 p)  (per axis), i.e. non-negative;
  4. argsort is stable (ties = points of one voxel, in index order); a voxel is represented to its NEIGHBOURS by its
     lowest-index point, and the centre tap of a submanifold convolution is the point itself.
"""
import math

import torch
import torch.nn.functional as F

ORDERS = ("z", "z-trans", "hilbert", "hilbert-trans")


# ------------------------------------------------------------------------------------------------ serialisation
def _interleave(a, b, c, depth):
    """bit i of a -> 3i+2, of b -> 3i+1, of c -> 3i (z_order.py:42-52)."""
    key = torch.zeros_like(a)
    for i in range(depth):
        m = 1 << i
        key = key | ((a & m) << (2 * i + 2)) | ((b & m) << (2 * i + 1)) | ((c & m) << (2 * i))
    return key


def z_order_code(grid, depth):
    """z_order.py:86-118: the LUT form equals the plain interleave of the low `depth` bits of x, y, z."""
    g = grid.long() & ((1 << depth) - 1)
    return _interleave(g[:, 0], g[:, 1], g[:, 2], depth)


def hilbert_code(grid, depth):
    """hilbert.py:93-190 (Skilling's transpose, vectorised over bit planes there; integer form here)."""
    X = [grid[:, 0].long().clone(), grid[:, 1].long().clone(), grid[:, 2].long().clone()]
    for bit in range(depth):  # most significant bit first
        Q = 1 << (depth - 1 - bit)
        P = Q - 1
        for d in range(3):
            on = (X[d] & Q) != 0
            X[0] = torch.where(on, X[0] ^ P, X[0])                  # bit on: invert the lower bits of axis 0
            t = torch.where(on, torch.zeros_like(X[0]), (X[0] ^ X[d]) & P)  # bit off: exchange lower bits with axis 0
            X[d] = X[d] ^ t
            X[0] = X[0] ^ t
    g = _interleave(X[0], X[1], X[2], depth)       # hilbert.py:170-171: [bit][dim] flattened, axis 0 most significant
    shift = 1
    while shift < 3 * depth:                       # gray2binary (hilbert.py:68-90): prefix xor from the top bit
        g = g ^ (g >> shift)
        shift *= 2
    return g


def encode(grid, batch, depth, order):
    """serialization/default.py:10-27."""
    if order == "z":
        code = z_order_code(grid, depth)
    elif order == "z-trans":
        code = z_order_code(grid[:, [1, 0, 2]], depth)
    elif order == "hilbert":
        code = hilbert_code(grid, depth)
    elif order == "hilbert-trans":
        code = hilbert_code(grid[:, [1, 0, 2]], depth)
    else:
        raise NotImplementedError(order)
    if batch is not None:
        code = batch.long() << (depth * 3) | code
    return code


def stable_argsort(code):
    return torch.sort(code, dim=-1, stable=True)[1]


def invert_order(order):
    inv = torch.zeros_like(order)
    return inv.scatter_(1, order, torch.arange(order.shape[1]).repeat(order.shape[0], 1))


def serialization(grid_coord, batch, orders=ORDERS):
    """pointtransformer_v3.py:81-145 without the shuffle -> (code [k,n], order [k,n], inverse [k,n], depth)."""
    depth = int(grid_coord.max()).bit_length()
    code = torch.stack([encode(grid_coord, batch, depth, o) for o in orders])
    order = stable_argsort(code)
    return code, order, invert_order(order), depth


# ------------------------------------------------------------------------------------------ submanifold convolution
def voxel_table(grid_coord, batch):
    """(batch, x, y, z) -> lowest row index holding that voxel."""
    table = {}
    for i, (b, g) in enumerate(zip(batch.tolist(), grid_coord.tolist())):
        table.setdefault((b, g[0], g[1], g[2]), i)
    return table


def neighbor_table(grid_coord, batch, ksize):
    """[n, k^3] int64: row gathered by tap (a, b, c) -> offset (a, b, c) - k//2 on (x, y, z); -1 where no point."""
    table = voxel_table(grid_coord, batch)
    n, r = grid_coord.shape[0], ksize // 2
    nbr = torch.full((n, ksize ** 3), -1, dtype=torch.long)
    gl, bl = grid_coord.tolist(), batch.tolist()
    for i in range(n):
        x, y, z = gl[i]
        t = 0
        for a in range(-r, r + 1):
            for b in range(-r, r + 1):
                for c in range(-r, r + 1):
                    nbr[i, t] = i if (a == 0 and b == 0 and c == 0) else table.get((bl[i], x + a, y + b, z + c), -1)
                    t += 1
    return nbr


def subm_conv3d(feat, nbr, weight, bias=None):
    """spconv SubMConv3d: out[i] = bias + sum_t W[:, t, :] feat[nbr[i, t]], weight [C_out, k, k, k, C_in]."""
    co, ci = weight.shape[0], weight.shape[-1]
    w = weight.reshape(co, -1, ci)
    out = torch.zeros(feat.shape[0], co, dtype=feat.dtype)
    for t in range(w.shape[1]):
        rows = (nbr[:, t] >= 0).nonzero()[:, 0]
        if rows.numel():
            out[rows] += feat[nbr[rows, t]] @ w[:, t, :].t()
    return out if bias is None else out + bias


# ----------------------------------------------------------------------------------------------------- attention
def patch_layout(count, patch_max):
    """pointtransformer_v3.py:392-447 for one cloud: (K, pad [n_pad] -> sorted position, unpad [n] -> padded position)."""
    K = min(count, patch_max)
    n_pad = (count + K - 1) // K * K
    pad = torch.arange(n_pad)
    if n_pad != count:  # only when count > K: the tail of the last patch borrows from the patch before it
        r = count % K
        pad[n_pad - K + r:] = pad[n_pad - 2 * K + r:n_pad - K]
    return K, pad, torch.arange(count)


def serialized_attention(p, prefix, feat, order, inverse, heads, patch_max):
    """pointtransformer_v3.py:449-499 (enable_flash=False, no RPE, upcasts are no-ops in fp32)."""
    n, C = feat.shape
    K, pad, unpad = patch_layout(n, patch_max)
    qkv = F.linear(feat, p[prefix + "qkv.weight"], p[prefix + "qkv.bias"])[order[pad]]
    q, k, v = qkv.reshape(-1, K, 3, heads, C // heads).permute(2, 0, 3, 1, 4).unbind(0)
    scale = (C // heads) ** -0.5
    attn = torch.softmax((q * scale) @ k.transpose(-2, -1), dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(-1, C)[unpad[inverse]]
    return F.linear(out, p[prefix + "proj.weight"], p[prefix + "proj.bias"])


def _bn(p, prefix, x, eps=1e-3):
    """BatchNorm1d(eps=1e-3) in eval mode (pointtransformer_v3.py:857)."""
    return F.batch_norm(x, p[prefix + "running_mean"], p[prefix + "running_var"], p[prefix + "weight"], p[prefix + "bias"],
                        False, 0.0, eps)


def _ln(p, prefix, x):
    return F.layer_norm(x, x.shape[-1:], p[prefix + "weight"], p[prefix + "bias"], 1e-5)


def block(p, prefix, feat, nbr, order, inverse, heads, patch_max, conv_in=None):
    """pointtransformer_v3.py:595-615 (pre_norm, drop_path = identity in eval).  `conv_in`: the features the sparse
    tensor holds when they differ from `feat` (first block after an unpooling, see unpooling())."""
    x = subm_conv3d(feat if conv_in is None else conv_in, nbr, p[prefix + "cpe.0.weight"], p[prefix + "cpe.0.bias"])
    x = _ln(p, prefix + "cpe.2.", F.linear(x, p[prefix + "cpe.1.weight"], p[prefix + "cpe.1.bias"]))
    feat = feat + x
    x = serialized_attention(p, prefix + "attn.", _ln(p, prefix + "norm1.0.", feat), order, inverse, heads, patch_max)
    feat = feat + x
    x = _ln(p, prefix + "norm2.0.", feat)
    x = F.linear(F.gelu(F.linear(x, p[prefix + "mlp.0.fc1.weight"], p[prefix + "mlp.0.fc1.bias"])),
                 p[prefix + "mlp.0.fc2.weight"], p[prefix + "mlp.0.fc2.bias"])
    return feat + x


# ------------------------------------------------------------------------------------------------------- pooling
def pooling(p, prefix, level):
    """pointtransformer_v3.py:648-721 (stride 2, reduce='max', no shuffle) on one cloud.
    level: dict(feat, grid, code [k,n], order, inverse, depth).  Returns (child level, cluster [n])."""
    shift = 1 if level["depth"] >= 1 else 0  # :649-651
    code = level["code"] >> (3 * shift)
    order0 = level["order"][0]
    sorted_parent = code[0][order0]
    first = torch.ones_like(sorted_parent, dtype=torch.bool)
    first[1:] = sorted_parent[1:] != sorted_parent[:-1]
    cid = torch.cumsum(first.long(), 0) - 1          # cluster of each sorted position == torch.unique's inverse
    cluster = torch.empty_like(cid)
    cluster[order0] = cid
    head = order0[first]                             # a member of each cluster (all members agree on what is read from it)
    proj = F.linear(level["feat"], p[prefix + "proj.weight"], p[prefix + "proj.bias"])
    m = int(cid[-1]) + 1
    pooled = torch.full((m, proj.shape[1]), -math.inf, dtype=proj.dtype).scatter_reduce(
        0, cluster[:, None].expand_as(proj), proj, reduce="amax")
    feat = F.gelu(_bn(p, prefix + "norm.0.", pooled))
    ccode = code[:, head]
    corder = stable_argsort(ccode)
    child = dict(feat=feat, grid=level["grid"][head] >> shift, code=ccode, order=corder, inverse=invert_order(corder),
                 depth=level["depth"] - shift)
    return child, cluster


def unpooling(p, prefix, child_feat, parent_feat, cluster):
    """pointtransformer_v3.py:748-759 with BatchNorm + GELU on both branches (:938-947) -> (feat, skip branch).
    The parent's sparse tensor is refreshed by proj_skip (PointSequential, :250-255) but NOT by the sum at :755, so the
    convolution of the next block reads the skip branch alone while its shortcut is the sum (:596-598)."""
    a = F.gelu(_bn(p, prefix + "proj.1.", F.linear(child_feat, p[prefix + "proj.0.weight"], p[prefix + "proj.0.bias"])))
    b = F.gelu(_bn(p, prefix + "proj_skip.1.", F.linear(parent_feat, p[prefix + "proj_skip.0.weight"],
                                                          p[prefix + "proj_skip.0.bias"])))
    return b + a[cluster], b


# -------------------------------------------------------------------------------------------------------- network
def ptv3_cloud(p, prefix, grid_coord, feat, cfg):
    """PointTransformerV3.forward (pointtransformer_v3.py:975-991) on ONE cloud with non-negative grid_coord [n,3]
    and feat [n, C_in]; cfg: enc_depths, enc_num_head, enc_patch_size, dec_depths, dec_num_head, dec_patch_size.
    -> feat [n, dec_channels[0]] in the input's point order."""
    batch = torch.zeros(grid_coord.shape[0], dtype=torch.long)
    code, order, inverse, depth = serialization(grid_coord, batch)
    level = dict(grid=grid_coord.long(), code=code, order=order, inverse=inverse, depth=depth)
    zeros = lambda lv: torch.zeros(lv["grid"].shape[0], dtype=torch.long)
    x = subm_conv3d(feat, neighbor_table(level["grid"], batch, 5), p[prefix + "embedding.stem.conv.weight"])
    level["feat"] = F.gelu(_bn(p, prefix + "embedding.stem.norm.", x))
    stack = []
    n_orders = len(ORDERS)
    for s, nblocks in enumerate(cfg["enc_depths"]):
        if s > 0:
            child, cluster = pooling(p, f"{prefix}enc.enc{s}.down.", level)
            stack.append((level, cluster))
            level = child
        level["nbr"] = neighbor_table(level["grid"], zeros(level), 3)
        for i in range(nblocks):
            k = i % n_orders
            level["feat"] = block(p, f"{prefix}enc.enc{s}.block{i}.", level["feat"], level["nbr"], level["order"][k],
                                  level["inverse"][k], cfg["enc_num_head"][s], cfg["enc_patch_size"][s])
    for s in reversed(range(len(cfg["dec_depths"]))):
        parent, cluster = stack.pop()
        parent["feat"], skip = unpooling(p, f"{prefix}dec.dec{s}.up.", level["feat"], parent["feat"], cluster)
        level = parent
        for i in range(cfg["dec_depths"][s]):
            k = i % n_orders
            level["feat"] = block(p, f"{prefix}dec.dec{s}.block{i}.", level["feat"], level["nbr"], level["order"][k],
                                  level["inverse"][k], cfg["dec_num_head"][s], cfg["dec_patch_size"][s],
                                  conv_in=skip if i == 0 else None)
    return level["feat"]


def frame_grid(points, grid_resolution=100):
    """Definition 3 of the header: floor(100 p) shifted to the frame's own origin (point_encoder.py:25-33)."""
    g = torch.floor(points * grid_resolution).long()
    return g - g.min(0)[0]


def encoder_forward(p, prefix, pts, feats, cfg):
    """PTv3Encoder.forward (point_encoder.py:25-40), frame by frame.  pts [B,N,3], feats [B,N,C] -> [B*N, C_out]."""
    return torch.cat([ptv3_cloud(p, prefix + "point_transformer.", frame_grid(pts[b]), feats[b], cfg)
                      for b in range(pts.shape[0])])


def point_refiner(p, prefix, point_features):
    """renderer.py:39-45,147: Linear-ReLU-Linear-ReLU-Linear -> offsets [.., 3]."""
    x = F.relu(F.linear(point_features, p[prefix + "0.weight"], p[prefix + "0.bias"]))
    x = F.relu(F.linear(x, p[prefix + "2.weight"], p[prefix + "2.bias"]))
    return F.linear(x, p[prefix + "4.weight"], p[prefix + "4.bias"])
