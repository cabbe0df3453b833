"""AudioTriplaneNet: audio tokens + SMPL-X/triplane tokens -> future tokens -> rendered frames.

Mirror of src/models/triplane_audio_net.py: same classes, constructor arguments, forward signature and 5-tuple
return (`rendered_images, gaussians, smpl_params, output_triplane_tokens, output_smpl_tokens`, :271), same
parameter names (`triplane_motion_encoder.conv_time`, `smplx_motion_encoder.{self_attn,mlp,norm1,norm2}`,
`transformer.*`).  Deterministic inference semantics (eval mode).

The autoregressive loop is inherently sequential (step t consumes step t-1's tokens, :224-253); per step the cost is
the 8-layer transformer over S = 2 * (3 R^2 + L) = 6304 tokens, whose self-attention runs on the MFMA kernel.
"""
import torch
import torch.nn as nn

from .transformer import Transformer1D_nn


class TriPlaneTemporalReducer(nn.Module):
    """:7-42.  Depthwise Conv3d with kernel (T,1,1) and no bias == a per-channel weighted sum over the T frames; the
    Conv3d module only holds the weight under its reference name, the sum is evaluated directly on the token layout
    (the reference permutes to NCDHW and back around the convolution)."""

    def __init__(self, C, time_steps):
        super().__init__()
        self.C, self.T, self.planes = C, time_steps, 3
        self.conv_time = nn.Conv3d(self.planes * C, self.planes * C, kernel_size=(self.T, 1, 1), stride=1,
                                   padding=(0, 0, 0), groups=self.planes * C, bias=False)

    def forward(self, x):
        """x [B,T,3,C,H,W] -> [B,1,3,C,H,W]"""
        B, T, P, C, H, W = x.shape
        assert P == self.planes and T == self.T and C == self.C, \
            f"Expected (B,{self.T},3,{self.C},H,W), got {tuple(x.shape)}"
        w = self.conv_time.weight.view(P, C, T)
        out = x[:, 0] * w[None, :, :, 0, None, None]
        for t in range(1, T):
            out = out + x[:, t] * w[None, :, :, t, None, None]
        return out.unsqueeze(1)

    def forward_tokens(self, tokens):
        """tokens [B,T,C,3*R*R] (reference token layout) -> [B,C,3*R*R]: the same sum without any rearrangement."""
        B, T, C, S = tokens.shape
        w = self.conv_time.weight.view(self.planes, C, T)                       # [plane, channel, t]
        w = w.permute(2, 1, 0).reshape(T, C, self.planes, 1).expand(T, C, self.planes, S // self.planes)
        w = w.reshape(T, C, S)
        out = tokens[:, 0] * w[0]
        for t in range(1, T):
            out = out + tokens[:, t] * w[t]
        return out


class SMPLXTemporalReducer(nn.Module):
    """:44-89: 2-token self-attention (nn.MultiheadAttention) + LN + MLP + LN, mean over time."""

    def __init__(self, C, time_steps):
        super().__init__()
        self.C, self.T = C, time_steps
        self.self_attn = nn.MultiheadAttention(embed_dim=C, num_heads=8, dropout=0.1, batch_first=True)
        self.mlp = nn.Sequential(nn.Linear(C, C * 2), nn.ReLU(), nn.Linear(C * 2, C))
        self.norm1 = nn.LayerNorm(C)
        self.norm2 = nn.LayerNorm(C)

    def forward(self, x):
        """x [B,T,C,S] -> [B,1,C,S]"""
        B, T, C, S = x.shape
        assert T == self.T and C == self.C, f"Expected (B,{self.T},{self.C},S), got {tuple(x.shape)}"
        x = x.permute(0, 3, 1, 2).reshape(B * S, T, C)
        attn_out, _ = self.self_attn(x, x, x, need_weights=False)
        x = self.norm1(x + attn_out)
        x = self.norm2(x + self.mlp(x))
        x = x.mean(dim=1, keepdim=True)
        return x.reshape(B, S, 1, C).permute(0, 2, 3, 1)


class AudioTriplaneNet(nn.Module):
    def __init__(self, cfg, renderer=None):
        super().__init__()
        self.cfg = cfg.model.triplane_audio_net
        self.T_input = self.cfg.triplane_input_frames
        self.T_output = self.cfg.triplane_output_frames
        self.triplane_motion_encoder = TriPlaneTemporalReducer(C=self.cfg.triplane_feature_dim,
                                                               time_steps=self.T_input)
        self.smplx_motion_encoder = SMPLXTemporalReducer(C=self.cfg.smpl_token_dim, time_steps=self.T_input)
        self.triplane_token_len = 3 * self.cfg.triplane_resolution * self.cfg.triplane_resolution
        self.smplx_token_len = self.cfg.smpl_token_len
        self.transformer = Transformer1D_nn(
            num_layers=self.cfg.transformer_layers, attention_head_dim=self.cfg.transformer_head_dim,
            in_channels=self.cfg.triplane_feature_dim, num_attention_heads=self.cfg.transformer_num_heads,
            cross_attention_dim=self.cfg.audio_feature_dim, norm_type="layer_norm",
            enable_memory_efficient_attention=False, gradient_checkpointing=True)
        self.renderer = renderer

    def generate_tokens(self, audio_features, input_triplane_tokens, smpl_tokens, num_steps=None):
        """The autoregressive part of forward (:181-266) -> (triplane tokens [B,T_out,C,3R^2], smpl tokens
        [B,T_out,D,L]).  `num_steps` (default T_output) lets a caller roll a longer clip from one call."""
        steps = self.T_output if num_steps is None else num_steps
        if audio_features.shape[1] < steps:
            raise ValueError(f"need {steps} audio tokens, got {audio_features.shape[1]}")
        L, S3 = self.smplx_token_len, self.triplane_token_len
        motion_tokens = self.triplane_motion_encoder.forward_tokens(input_triplane_tokens)
        smplx_motion = self.smplx_motion_encoder(smpl_tokens).squeeze(1)
        last_tri, last_smpl = input_triplane_tokens[:, -1], smpl_tokens[:, -1]
        query = torch.cat([motion_tokens, smplx_motion, last_tri, last_smpl], dim=-1)
        out_tri, out_smpl = [], []
        for t in range(steps):
            out = self.transformer(query, audio_features[:, t:t + 1])
            smpl = out[:, :, -L:]
            tri = out[:, :, -S3 - L:-L]
            # note the reference's frame order: triplane [pred, last] (:240) but smplx [last, pred] (:246)
            tri_motion = self.triplane_motion_encoder.forward_tokens(torch.stack([tri, last_tri], dim=1))
            smpl_motion = self.smplx_motion_encoder(torch.stack([last_smpl, smpl], dim=1)).squeeze(1)
            query = torch.cat([tri_motion, smpl_motion, tri, smpl], dim=-1)
            last_tri, last_smpl = tri, smpl
            out_tri.append(tri)
            out_smpl.append(smpl)
        return torch.stack(out_tri, dim=1), torch.stack(out_smpl, dim=1)

    def forward(self, audio_features, input_triplane_tokens, ref_image_features, cam_params, smpl_tokens):
        """Args as :157-167 (ref_image_features is accepted and unused, as in the reference)."""
        output_triplane_tokens, output_smpl_tokens = self.generate_tokens(audio_features, input_triplane_tokens,
                                                                          smpl_tokens)
        rendered_images, gaussians, smpl_params = self.renderer(output_triplane_tokens, cam_params,
                                                                output_smpl_tokens)
        return rendered_images, gaussians, smpl_params, output_triplane_tokens, output_smpl_tokens
