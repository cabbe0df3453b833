"""Configuration objects with the reference's field names.

The reference flattens its YAML tree into one namespace (src/configs/config_loader.py:190-234) and hands it to
`Renderer(cfg)` / `SMPLXDecoder(cfg)`, while `AudioTriplaneNet(cfg)` reads `cfg.model.triplane_audio_net.*`
(src/models/triplane_audio_net.py:94).  Any object with these attributes works (an OmegaConf node from the
reference included); these dataclasses are the Lightning/omegaconf-free way to build one.  Defaults are the
reference's YAML values (src/configs/model/*.yaml, src/configs/datasets/ted_speech.yaml) except for the two
switches SURVEY.md section 8(f) defers: `upsample_triplane` and the PTv3 point refiner are off.
"""
from dataclasses import dataclass, field
from types import SimpleNamespace
from typing import Optional, Tuple

import yaml


@dataclass
class RendererConfig:
    # src/configs/model/triplane_net.yaml:3-16
    triplane_resolution: int = 32
    triplane_feature_dim: int = 256
    radius: float = 1.4
    smplx_model_path: Optional[str] = None
    smpl_token_len: int = 80
    smpl_token_dim: int = 256
    # src/configs/datasets/ted_speech.yaml:14-17
    flat_hand_mean: bool = True
    num_expression_coeffs: int = 10
    image_size: Tuple[int, int] = (512, 512)  # (H, W); BASELINE configs use 512x512
    # src/configs/model/renderer.yaml:10-22
    upsample_triplane: bool = False   # reference default true: SURVEY 8(f) next-row
    num_upsample_blocks: int = 4
    upsample_windows: bool = True     # upsample only the plane regions the body's points can sample (renderer.py, TriplaneUpsampler)
    upsample_frames_per_pass: int = 8    # Renderer.forward splits longer calls (upsampled planes: 805 MB per frame)
    upsample_window_margin: float = 0.05  # metres kept free around the points for the refiner's offsets (checked; falls back)
    densify_smplx_verts: bool = True
    subdivide_steps: int = 0          # 0 -> 10 000 sampled vertices (BASELINE), reference default 2 -> 30 000
    predict_smplx_params: bool = True
    no_point_refiner: bool = True     # reference default false: PTv3 refiner (point_transformer.py); an untrained
                                      # refiner's last layer is zero (renderer.py:46-47), i.e. offsets == 0
    # src/configs/model/ptv3_encoder.yaml:5-20 (input_dim = 3 * triplane_feature_dim when None)
    input_dim: Optional[int] = None
    stride: Tuple[int, ...] = (2, 2, 2, 2)
    enc_channels: Tuple[int, ...] = (32, 64, 128, 256, 512)
    enc_depths: Tuple[int, ...] = (2, 2, 2, 6, 2)
    dec_channels: Tuple[int, ...] = (256, 128, 256, 512)
    dec_depths: Tuple[int, ...] = (2, 2, 2, 2)
    enc_num_head: Tuple[int, ...] = (2, 4, 8, 16, 32)
    dec_num_head: Tuple[int, ...] = (4, 4, 8, 16)
    enc_patch_size: Tuple[int, ...] = (512, 512, 512, 512, 512)
    dec_patch_size: Tuple[int, ...] = (512, 512, 512, 512)
    enable_flash: bool = False
    refiner_clouds_per_pass: int = 32  # frames refined together: the network is ~900 launches per pass whatever the number of
    #                                   clouds, so larger passes cost less per frame (10 k points: 8 -> 2.26, 16 -> 1.85,
    #                                   32 -> 1.62 ms per frame); frames are independent, the split changes nothing
    refiner_points_per_pass: int = 320_000  # ... but a pass holds at most this many points (working set ~0.4 MB per 1 k points)
    use_gaussian_splatting: bool = True
    gaussian_feature_dim: int = 256
    rgb: bool = True
    sh_degree: int = 3
    device: str = "cuda"
    # deterministic-inference additions (SURVEY.md "Hard parts"): the reference re-draws the vertex subset with
    # torch.randperm on every forward (renderer.py:287); here it is drawn once from this seed.
    num_gaussians: Optional[int] = None  # None: the reference's SUBDEVIDE_VERTS[subdivide_steps]
    subset_seed: int = 42
    subset_order: str = "random"      # "spatial": the same subset, stored along a Z-order curve of the rest pose
    body_seed: int = 42               # seed of the synthetic body used when smplx_model_path is absent
    pipeline_chunks: int = 1          # >1: frame groups on separate HIP streams (measured slower on MI355X: 1.53 -> 1.79 ms)


@dataclass
class Stage1Config(RendererConfig):
    """The flattened model config `TriplaneGaussianAvatar` hands to its parts (config_loader.py:190-234): the
    renderer's fields plus src/configs/model/triplane_net.yaml:10-30 and sapiens_encoder.yaml:3."""
    smplx_transformer_layers: int = 4
    smplx_transformer_head_dim: int = 64
    smplx_transformer_num_heads: int = 8
    cross_transformer_layers: int = 8
    cross_transformer_head_dim: int = 64
    cross_transformer_num_heads: int = 8
    image_feature_dim: int = 1536
    sample_feature: bool = True
    upsample_factor: int = 3


@dataclass
class AudioNetConfig:
    # src/configs/model/triplane_audio_net.yaml:3-14
    triplane_input_frames: int = 2
    triplane_output_frames: int = 6
    triplane_feature_dim: int = 256
    triplane_resolution: int = 32
    smpl_token_len: int = 80
    smpl_token_dim: int = 256
    transformer_layers: int = 8
    transformer_head_dim: int = 64
    transformer_num_heads: int = 8
    audio_feature_dim: int = 768


@dataclass
class ModelConfig:
    """`cfg` as AudioTriplaneNet expects it: cfg.model.triplane_audio_net.* (triplane_audio_net.py:94)."""
    triplane_audio_net: AudioNetConfig = field(default_factory=AudioNetConfig)
    renderer: RendererConfig = field(default_factory=RendererConfig)

    @property
    def model(self):
        return self


def load_yaml(path: str) -> SimpleNamespace:
    """Plain YAML -> nested attribute namespace (PyYAML safe loader; no `${...}` interpolation, no eval)."""
    with open(path) as f:
        data = yaml.safe_load(f)

    def ns(x):
        if isinstance(x, dict):
            return SimpleNamespace(**{k: ns(v) for k, v in x.items()})
        if isinstance(x, list):
            return [ns(v) for v in x]
        return x

    return ns(data)
