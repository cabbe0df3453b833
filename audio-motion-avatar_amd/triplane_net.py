"""Stage-1 identity encoder: reference image tokens + SMPL-X -> fused triplane / SMPL-X tokens -> rendered frame.

Mirror of `TriplaneGaussianAvatar` (src/models/lightning_model_wrapper.py:25-53) and of src/models/triplane_net.py
(`ResnetBlockFC` :16-58, `SMPLXTriplaneEncoder` :66-352, `FeatureFusionNetwork` :355-418), src/models/tokenizers.py
(`TriplaneLearnablePositionalEmbedding`) and src/models/image_feature.py:257-275 (`ImageFeature`): same classes,
constructor arguments, forward signatures, return tuples and parameter names, so the `triplane_gaussian.*` keys of a
reference checkpoint load.  SURVEY.md section 8(f) row 3 -- a "next row": it runs once per identity, so it is built for
correctness and determinism, on library kernels except where the reference's own operators are undefined on a GPU:

  * the Sapiens-1B image encoder (`sapiens_encoder`, a TorchScript file that is not available offline) is NOT part of
    this module: `forward` takes its output (`image_tokens [B,T,4096,1536]`) as an extra argument and raises when it is
    missing;
  * `torch_scatter.scatter_max / scatter_mean` (atomics: sum order undefined) -> deterministic segment reductions over
    points stably sorted by cell (csrc/splat.hip);
  * `points_projection` (pytorch3d point rasterizer + an index_put with duplicate indices: which pixel a point
    receives is undefined) -> z-buffer kernel with a fixed rule: the last pixel in (y, x) order that the point wins
    (what a sequential index_put does);
  * cross-attention to the 4096 image tokens -> the library's fused attention; self-attention -> the MFMA kernel.

`densify_smplx_verts` here means "append the face centres" (:266-272), not the Renderer's subdivision.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._lib import AmavError
from .body_model import BodyModel
from .renderer import Renderer
from .smplx_decoder import SMPLXDecoder
from .transformer import Transformer1D_nn

POINT_RADIUS_NDC = 0.0075  # graphic_utils.py:280 (pytorch3d NDC: the shorter image side spans [-1, 1])


class ResnetBlockFC(nn.Module):
    """triplane_net.py:16-58."""

    def __init__(self, size_in, size_out=None, size_h=None):
        super().__init__()
        size_out = size_in if size_out is None else size_out
        size_h = min(size_in, size_out) if size_h is None else size_h
        self.size_in, self.size_h, self.size_out = size_in, size_h, size_out
        self.fc_0 = nn.Linear(size_in, size_h)
        self.fc_1 = nn.Linear(size_h, size_out)
        self.actvn = nn.ReLU()
        self.shortcut = None if size_in == size_out else nn.Linear(size_in, size_out, bias=False)
        nn.init.zeros_(self.fc_1.weight)

    def forward(self, x):
        net = self.fc_0(self.actvn(x))
        dx = self.fc_1(self.actvn(net))
        return (x if self.shortcut is None else self.shortcut(x)) + dx


class TriplaneLearnablePositionalEmbedding(nn.Module):
    """tokenizers.py: learnable [3,C,R,R] embedding added to the planes, flattened to tokens `B Ct (Np Hp Wp)`."""

    def __init__(self, num_channels=1024, plane_size=32):
        super().__init__()
        self.plane_size, self.num_channels = plane_size, num_channels
        self.embeddings = nn.Parameter(torch.randn(3, num_channels, plane_size, plane_size) / math.sqrt(num_channels))

    def forward(self, batch_size, cond_embeddings=None):
        e = self.embeddings.unsqueeze(0).expand(batch_size, -1, -1, -1, -1)
        if cond_embeddings is not None:
            e = e + cond_embeddings
        return e.permute(0, 2, 1, 3, 4).reshape(batch_size, self.num_channels, -1)

    def detokenize(self, tokens):
        B, Ct, Nt = tokens.shape
        assert Nt == self.plane_size ** 2 * 3 and Ct == self.num_channels
        return tokens.reshape(B, Ct, 3, self.plane_size, self.plane_size).permute(0, 2, 1, 3, 4)


class ImageFeature(nn.Module):
    """image_feature.py:257-275: Linear(1536 -> 125) on the 64x64 token grid, bilinear resize to the image, cat RGB."""

    def __init__(self, image_feature_dim=1536):
        super().__init__()
        self.feature_reducer = nn.Linear(image_feature_dim, 128 - 3)

    def forward(self, rgb, feature):
        B, Nv, Nt, C = feature.shape
        H, W = rgb.shape[-2:]
        side = int(round(Nt ** 0.5))
        f = self.feature_reducer(feature.reshape(B * Nv * Nt, C)).reshape(B * Nv, side, side, -1).permute(0, 3, 1, 2)
        f = F.interpolate(f.contiguous(), size=(H, W), mode="bilinear", align_corners=False)
        return torch.cat([rgb.reshape(B * Nv, *rgb.shape[2:]), f], dim=1).reshape(B, Nv, -1, H, W)


class SMPLXTriplaneEncoder(nn.Module):
    """triplane_net.py:66-352: a posed SMPL-X mesh (+ per-vertex embedding, + image features under the projected
    vertices) -> PointNet with triplane-cell max pooling -> per-cell mean -> geometry triplanes [B,T,3,C,R,R]."""

    def __init__(self, cfg, smpl_decoder=None):
        super().__init__()
        self.cfg = cfg
        self.triplane_resolution = cfg.triplane_resolution
        self.feature_dim = C = cfg.triplane_feature_dim
        self.smplx_model = self.init_smplx_model()
        faces = torch.as_tensor(self.smplx_model.faces.astype("int64"))
        self.register_buffer("_faces", faces, persistent=False)
        self.num_verts = self.smplx_model.num_verts + (faces.shape[0] if cfg.densify_smplx_verts else 0)
        self.fc_pos = nn.Linear(3 + C, 2 * C)
        self.blocks = nn.ModuleList([ResnetBlockFC(2 * C, C) for _ in range(3)])
        self.fc_c = nn.Linear(C, C)
        self.vertex_emb = nn.Embedding(self.num_verts, C // 2 if cfg.sample_feature else C)
        if getattr(cfg, "upsample_triplane", False):
            raise NotImplementedError("stage 1 with upsample_triplane (TriplaneDownsampler, triplane_net.py:432-451) "
                                      "is not built; the reference's triplane_net.yaml has it off")
        if cfg.predict_smplx_params:
            self.smpl_token_len, self.smpl_token_dim = cfg.smpl_token_len, cfg.smpl_token_dim
            self.smpl_tokens = nn.Parameter(torch.randn(self.smpl_token_dim, self.smpl_token_len))
            self.cross_attn = Transformer1D_nn(
                num_layers=cfg.smplx_transformer_layers, attention_head_dim=cfg.smplx_transformer_head_dim,
                in_channels=self.smpl_token_dim, num_attention_heads=cfg.smplx_transformer_num_heads,
                cross_attention_dim=cfg.image_feature_dim, norm_type="layer_norm")
            self.smpl_decoder = smpl_decoder
        self.actvn = nn.ReLU()

    def init_smplx_model(self):
        return BodyModel.create(getattr(self.cfg, "smplx_model_path", None), device=self.cfg.device, num_betas=10,
                                num_expression_coeffs=self.cfg.num_expression_coeffs,
                                flat_hand_mean=self.cfg.flat_hand_mean, seed=getattr(self.cfg, "body_seed", 42))

    # ---- :209-224
    def smpl_predictor(self, image_features):
        B, T, S, C = image_features.shape
        query = self.smpl_tokens.unsqueeze(0).repeat(B * T, 1, 1)
        tokens = self.cross_attn(query, image_features.reshape(B * T, S, C))
        params = self.smpl_decoder(tokens)
        for key, v in list(params.items()):
            params[key] = v.reshape(B, T, *v.shape[1:]) if key in ("body_pose", "left_hand_pose", "right_hand_pose") \
                else v.reshape(B, T, -1)
        return params, tokens

    # ---- :266-274 (LBS, then the face centres appended)
    def get_smplx_verts(self, smpl_params):
        B, T = smpl_params["global_orient"].shape[:2]
        r = lambda k: smpl_params[k].reshape(B * T, -1)
        vertices = self.smplx_model(global_orient=r("global_orient"), body_pose=r("body_pose"), betas=r("betas"),
                                    left_hand_pose=r("left_hand_pose"), right_hand_pose=r("right_hand_pose"),
                                    jaw_pose=r("jaw_pose"), leye_pose=r("leye_pose"), reye_pose=r("reye_pose"),
                                    expression=r("expression")).vertices
        if not self.cfg.densify_smplx_verts:
            return vertices
        centers = vertices[:, self._faces].mean(dim=2)
        return torch.cat([vertices, centers], dim=1)

    def cell_indices(self, verts):
        """:163-183: clamp to the cube, normalise to [0,1), cell = x + R * y for (x,y), (x,z), (y,z) -> int32 [BT,3,N]."""
        R, rad = self.triplane_resolution, self.cfg.radius
        pos = (torch.clamp(verts, -rad + 1e-6, rad - 1e-6) + rad) / (2 * rad)
        cells = []
        for a, b in ((0, 1), (0, 2), (1, 2)):
            x = (pos[..., [a, b]] * R).long()
            cells.append(torch.clamp(x[..., 0] + R * x[..., 1], 0, R * R - 1))
        return torch.stack(cells, dim=1).to(torch.int32)

    def forward(self, cam_params, img_tokens, smpl_params_gt=None, img=None):
        B, T, S, C = img_tokens.shape
        pred_smpl_params = smpl_tokens = None
        if self.cfg.predict_smplx_params:
            pred_smpl_params, smpl_tokens = self.smpl_predictor(img_tokens)
        smpl_params = smpl_params_gt if smpl_params_gt is not None else pred_smpl_params
        if smpl_params is None:
            raise AmavError("SMPLXTriplaneEncoder: no SMPL-X parameters (predict_smplx_params off, no smpl_params_gt)")
        verts = self.get_smplx_verts(smpl_params)                                    # [BT, N, 3]
        verts_emb = self.vertex_emb.weight.unsqueeze(0).expand(verts.shape[0], -1, -1)
        if self.cfg.sample_feature:                                                  # :139-157
            Himg, Wimg = img.shape[-2:]
            pts = verts + smpl_params["transl"].reshape(B * T, 1, 3)
            sampled = ops.points_project(pts.contiguous(), cam_params["extrinsic"].reshape(B * T, 4, 4).float(),
                                         cam_params["intrinsic"].reshape(B * T, 3, 3).float(),
                                         img.reshape(B * T, *img.shape[2:]).float(),
                                         POINT_RADIUS_NDC * min(Himg, Wimg) / 2.0)
            verts_feat = torch.cat([verts_emb, sampled], dim=-1)
        else:
            verts_feat = verts_emb
        net = self.blocks[0](self.fc_pos(torch.cat([verts, verts_feat], dim=-1)))
        cell_of = self.cell_indices(verts)
        cells = self.triplane_resolution ** 2
        segments = ops.cell_segments(cell_of, cells)
        for block in self.blocks[1:]:                                                # :185-188
            pooled = ops.cell_pool_max(net, cell_of, cells, segments)
            net = block(torch.cat([net, pooled], dim=2))
        c = self.fc_c(net)
        planes = [ops.cell_splat_mean(c, cell_of[:, p].contiguous(), cells,
                                      (segments[0][:, p].contiguous(), segments[1][:, p].contiguous()))
                  for p in range(3)]
        R = self.triplane_resolution
        smplx_triplanes = torch.stack(planes, dim=1).view(B, T, 3, -1, R, R)
        return smplx_triplanes, smpl_tokens, pred_smpl_params


class FeatureFusionNetwork(nn.Module):
    """triplane_net.py:355-418: geometry planes + positional embedding, SMPL-X tokens appended, 8 transformer layers
    with cross-attention to the image tokens; split back."""

    def __init__(self, cfg, feature_dim=64):
        super().__init__()
        self.cfg = cfg
        self.triplane_resolution, self.triplane_feature_dim = cfg.triplane_resolution, cfg.triplane_feature_dim
        self.triplane_tokenizer_geometry = TriplaneLearnablePositionalEmbedding(self.triplane_feature_dim,
                                                                                self.triplane_resolution)
        self.transformer_cross = Transformer1D_nn(
            num_layers=cfg.cross_transformer_layers, attention_head_dim=cfg.cross_transformer_head_dim,
            in_channels=self.triplane_feature_dim, num_attention_heads=cfg.cross_transformer_num_heads,
            cross_attention_dim=1536, norm_type="layer_norm")  # hard-coded in the reference too (:326)

    def forward(self, geometry_triplane, image_features, smpl_tokens):
        B, T, _, C, H, W = geometry_triplane.shape
        geo = geometry_triplane.reshape(B * T, 3, C, H, W)
        img = image_features.reshape(B * T, *image_features.shape[2:])
        geo_tokens = self.triplane_tokenizer_geometry(batch_size=B * T, cond_embeddings=geo)
        combined = torch.cat([geo_tokens, smpl_tokens], dim=2)
        out = self.transformer_cross(combined, img)
        tokens, smpl_out = torch.split(out, [geo_tokens.shape[2], smpl_tokens.shape[2]], dim=2)
        return tokens.reshape(B, T, *tokens.shape[1:]), smpl_out.reshape(B, T, *smpl_out.shape[1:])


class TriplaneGaussianAvatar(nn.Module):
    """lightning_model_wrapper.py:25-53.  `cfg`: the flattened model config (config.RendererConfig + the stage-1
    fields below, or any object with those attributes).  The Sapiens encoder is external: pass its tokens."""

    def __init__(self, cfg=None):
        super().__init__()
        self.cfg = cfg
        self.sapiens_encoder = None  # TorchScript Sapiens-1B in the reference (lightning_model_wrapper.py:33)
        self.image_feature = ImageFeature(getattr(cfg, "image_feature_dim", 1536))
        self.smplx_decoder = SMPLXDecoder(cfg)
        self.smplx_triplane_encoder = SMPLXTriplaneEncoder(cfg, self.smplx_decoder)
        self.fusion_network = FeatureFusionNetwork(cfg)
        self.renderer = Renderer(cfg, self.smplx_decoder)
        self.to(cfg.device)

    def forward(self, img, smpl_params_gt, cam_params, image_tokens=None):
        """img [B,T,3,H,W], smpl_params_gt dict of [B,T,...] (or None), cam_params dict -> the reference's 7-tuple
        (rendered_images, gaussians, fused_triplane_tokens, image_tokens, pred_smpl_1, pred_smpl_2, smpl_tokens)."""
        B, T = img.shape[:2]
        if image_tokens is None:
            if self.sapiens_encoder is None:
                raise AmavError("TriplaneGaussianAvatar.forward: the Sapiens image encoder is not part of this build "
                                "(its TorchScript checkpoint is not available offline); pass image_tokens "
                                "[B,T,4096,image_feature_dim]")
            image_tokens = self.sapiens_encoder(img.reshape(B * T, *img.shape[2:]))
        image_tokens = image_tokens.reshape(B, T, -1, image_tokens.shape[-1])
        image_features = self.image_feature(img, image_tokens)
        smplx_triplane, smpl_tokens, pred_smpl_1 = self.smplx_triplane_encoder(cam_params, image_tokens, smpl_params_gt,
                                                                              image_features)
        fused_tokens, smpl_tokens = self.fusion_network(smplx_triplane, image_tokens, smpl_tokens)
        rendered_images, gaussians, pred_smpl_2 = self.renderer(fused_tokens, cam_params, smpl_tokens, smpl_params_gt)
        return rendered_images, gaussians, fused_tokens, image_tokens, pred_smpl_1, pred_smpl_2, smpl_tokens
