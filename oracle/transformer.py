"""Transformer1D_nn and the audio-driven token generator, restated on CPU (functional, weights by reference name).

Follows src/models/transformers.py:912-1074 (Transformer1D_nn), :140-399 (BasicTransformerBlock, layer_norm
variant), :402-452,484-508 (FeedForward / GEGLU), the diffusers `Attention` configuration used there
(transformers.py:226-234,250-260; diffusers is absent and unpinned -> PARITY UNPINNED) and
src/models/triplane_audio_net.py:7-271 (temporal reducers + AudioTriplaneNet.forward up to the renderer call).
Deterministic eval-mode semantics (dropout off).  Test infrastructure only.
"""
import einops
import torch
import torch.nn.functional as F


def attention(p, prefix, hidden, context=None, heads=8):
    """diffusers Attention (SDPA processor): q/k/v without bias, out with bias, scale 1/sqrt(dim_head)."""
    ctx = hidden if context is None else context
    q = F.linear(hidden, p[prefix + "to_q.weight"])
    k = F.linear(ctx, p[prefix + "to_k.weight"])
    v = F.linear(ctx, p[prefix + "to_v.weight"])
    B, S, inner = q.shape
    d = inner // heads
    sp = lambda t: t.view(B, -1, heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v), attn_mask=None, dropout_p=0.0, is_causal=False)
    o = o.transpose(1, 2).reshape(B, S, inner)
    return F.linear(o, p[prefix + "to_out.0.weight"], p[prefix + "to_out.0.bias"])


def feed_forward(p, prefix, x):
    """GEGLU (first half = value, second half = gate, exact-erf GELU) then Linear (transformers.py:438-452,497-508)."""
    h = F.linear(x, p[prefix + "net.0.proj.weight"], p[prefix + "net.0.proj.bias"])
    val, gate = h.chunk(2, dim=-1)
    return F.linear(val * F.gelu(gate), p[prefix + "net.2.weight"], p[prefix + "net.2.bias"])


def transformer_block(p, prefix, x, enc, heads=8):
    """transformers.py:292-399, norm_type='layer_norm'."""
    ln = lambda n, t: F.layer_norm(t, (t.shape[-1],), p[prefix + n + ".weight"], p[prefix + n + ".bias"], 1e-5)
    x = attention(p, prefix + "attn1.", ln("norm1", x), None, heads) + x
    x = attention(p, prefix + "attn2.", ln("norm2", x), enc, heads) + x
    x = feed_forward(p, prefix + "ff.", ln("norm3", x)) + x
    return x


def transformer1d(p, prefix, hidden, enc, num_layers=8, heads=8, groups=32):
    """transformers.py:1016-1074: [B,C,S] (+ [B,1,768]) -> [B,C,S]."""
    B, C, S = hidden.shape
    residual = hidden
    h = F.group_norm(hidden, groups, p[prefix + "norm.weight"], p[prefix + "norm.bias"], 1e-6)
    h = h.permute(0, 2, 1).reshape(B, S, C)
    h = F.linear(h, p[prefix + "proj_in.weight"], p[prefix + "proj_in.bias"])
    for i in range(num_layers):
        h = transformer_block(p, f"{prefix}transformer_blocks.{i}.", h, enc, heads)
    h = F.linear(h, p[prefix + "proj_out.weight"], p[prefix + "proj_out.bias"])
    h = h.reshape(B, S, C).permute(0, 2, 1).contiguous()
    return h + residual


def triplane_temporal_reducer(p, prefix, x):
    """triplane_audio_net.py:24-42: depthwise Conv3d kernel (T,1,1), no bias.  x [B,T,3,C,H,W] -> [B,1,3,C,H,W]."""
    B, T, P, C, H, W = x.shape
    xp = x.permute(0, 2, 3, 1, 4, 5).contiguous().view(B, P * C, T, H, W)
    out = F.conv3d(xp, p[prefix + "conv_time.weight"], None, 1, 0, 1, P * C)
    return out.view(B, P, C, 1, H, W).permute(0, 3, 1, 2, 4, 5).contiguous()


def smplx_temporal_reducer(p, prefix, x, heads=8):
    """triplane_audio_net.py:66-89 (eval mode: attention dropout off).  x [B,T,C,S] -> [B,1,C,S]."""
    B, T, C, S = x.shape
    x = einops.rearrange(x, "b t c s -> (b s) t c")
    xt = x.transpose(0, 1)  # MultiheadAttention(batch_first=True) works on [T, N, C] internally
    attn, _ = F.multi_head_attention_forward(
        xt, xt, xt, C, heads, p[prefix + "self_attn.in_proj_weight"], p[prefix + "self_attn.in_proj_bias"], None,
        None, False, 0.0, p[prefix + "self_attn.out_proj.weight"], p[prefix + "self_attn.out_proj.bias"],
        training=False, need_weights=False)
    attn = attn.transpose(0, 1)
    x = F.layer_norm(x + attn, (C,), p[prefix + "norm1.weight"], p[prefix + "norm1.bias"])
    m = F.linear(F.relu(F.linear(x, p[prefix + "mlp.0.weight"], p[prefix + "mlp.0.bias"])),
                 p[prefix + "mlp.2.weight"], p[prefix + "mlp.2.bias"])
    x = F.layer_norm(x + m, (C,), p[prefix + "norm2.weight"], p[prefix + "norm2.bias"])
    x = x.mean(dim=1, keepdim=True)
    return einops.rearrange(x, "(b s) t c -> b t c s", b=B)


def audio_triplane_tokens(p, audio_features, input_triplane_tokens, smpl_tokens, resolution=32, smpl_len=80,
                          t_output=6, num_layers=8, heads=8, prefix=""):
    """AudioTriplaneNet.forward up to the renderer call (triplane_audio_net.py:157-266).

    Returns (output_triplane_tokens [B,T_out,C,3R^2], output_smpl_tokens [B,T_out,D,L]).
    """
    R = resolution
    tri_len = 3 * R * R
    to_planes = lambda tok: einops.rearrange(tok, "b c (np h w) -> b np c h w", np=3, h=R, w=R)
    input_triplanes = einops.rearrange(input_triplane_tokens, "b t c (np h w) -> b t np c h w", np=3, h=R, w=R)
    motion = triplane_temporal_reducer(p, prefix + "triplane_motion_encoder.", input_triplanes).squeeze(1)
    motion_tokens = einops.rearrange(motion, "b np c h w -> b c (np h w)")
    smplx_motion = smplx_temporal_reducer(p, prefix + "smplx_motion_encoder.", smpl_tokens).squeeze(1)
    last_tri = input_triplane_tokens[:, -1]
    last_smpl = smpl_tokens[:, -1]
    query = torch.cat([motion_tokens, smplx_motion, last_tri, last_smpl], dim=-1)
    out_tri, out_smpl = [], []
    for t in range(t_output):
        out = transformer1d(p, prefix + "transformer.", query, audio_features[:, t:t + 1], num_layers, heads)
        smpl = out[:, :, -smpl_len:]
        tri = out[:, :, -tri_len - smpl_len:-smpl_len]
        pred_plane = to_planes(tri).unsqueeze(1)
        last_tri = out_tri[-1] if out_tri else last_tri
        last_plane = to_planes(last_tri).unsqueeze(1)
        last_smpl = out_smpl[-1] if out_smpl else last_smpl
        # note the order: triplane [pred, last] (:240) but smplx [last, pred] (:246)
        tri_motion = triplane_temporal_reducer(p, prefix + "triplane_motion_encoder.",
                                               torch.cat([pred_plane, last_plane], dim=1)).squeeze(1)
        tri_motion_tokens = einops.rearrange(tri_motion, "b np c h w -> b c (np h w)")
        smpl_motion = smplx_temporal_reducer(p, prefix + "smplx_motion_encoder.",
                                             torch.cat([last_smpl.unsqueeze(1), smpl.unsqueeze(1)], dim=1)).squeeze(1)
        query = torch.cat([tri_motion_tokens, smpl_motion, tri, smpl], dim=-1)
        out_tri.append(tri)
        out_smpl.append(smpl)
    return torch.stack(out_tri, dim=1), torch.stack(out_smpl, dim=1)
