"""Lightning-free caller of the hot path: the build's counterpart of `AudioDrivenTriplaneAvatarLightning`
(src/models/lightning_model_wrapper.py:392-657) and of the `main2.py --mode demo` loop (src/main2.py:123-340).

Only the inference I/O of the boundary is reproduced (SURVEY.md section 8b "caller harness"):

  * `predict_step`  : (reference tokens, audio + cameras) -> `audio_rendered_images [B,T_out,H,W,3]`
                      (lightning_model_wrapper.py:574-605,640).  The stage-1 encoder that turns reference IMAGES
                      into tokens (`self.triplane_gaussian`, :592-597) is out of scope: the
                      caller passes the tokens it produced.
  * `rollout`       : the demo's window chaining -- every window of T_out frames is generated from the last two
                      outputs of the previous one (main2.py:179-203: `triplanes, smplx_tokens = out[:, -2:]`).
  * `rollout_interleaved`: the demo's even / odd chains over a stride-2 clip, run as one batch of two and zipped
                      (main2.py:160-311).
  * `rollout_sharded`: the multi-GPU form: each rank rolls and renders its own contiguous block of windows
                      (segment-parallel, SURVEY section 8e option i; `mode="sequential"` = option ii: rank 0 runs the
                      one exact chain and hands out token blocks) and the clip is reassembled by an all-gather of
                      uint8 frames.
  * `load_reference_checkpoint`: `checkpoint['state_dict']` with the `audio_triplane.` / `triplane_gaussian.renderer.`
                      prefixes (main2.py:127-138, strict=False like the reference).

Deterministic inference semantics: modules run in eval mode (the reference's demo leaves them in .train(),
main2.py:140-141, which makes its own output non-reproducible).
"""
import torch
import torch.nn as nn

from . import ops
from .config import ModelConfig
from .dist import all_gather_frames, recv_frames, send_frames, shard_range
from .renderer import Renderer
from .smplx_decoder import SMPLXDecoder
from .triplane_audio_net import AudioTriplaneNet
from .tuning import use_tuned_gemms


class AudioDrivenAvatar(nn.Module):
    def __init__(self, cfg: ModelConfig = None):
        super().__init__()
        self.cfg = cfg or ModelConfig()
        rcfg = self.cfg.renderer
        use_tuned_gemms()  # library kernel selection for the transformer's fixed GEMM shapes (tuning.py)
        self.smpl_decoder = SMPLXDecoder(rcfg)
        # one Renderer shared by both stages, as in the reference (lightning_model_wrapper.py:405-406)
        self.renderer = Renderer(rcfg, smpl_decoder=self.smpl_decoder)
        self.audio_triplane = AudioTriplaneNet(self.cfg, renderer=self.renderer)
        self.to(rcfg.device).eval()

    # ---- checkpoint ----------------------------------------------------------------------------------------------
    def load_reference_checkpoint(self, path_or_state, strict=False):
        """Load the tensors of a reference checkpoint (`checkpoint['state_dict']` with the `audio_triplane.` /
        `triplane_gaussian.renderer.` prefixes, main2.py:127-138).  Uses torch.load(weights_only=True): nothing from
        the file is executed.

        Only keys of modules this Renderer was built without are dropped (`point_encoder.*` / `point_refiner.*` with
        cfg.no_point_refiner, `triplane_upsampler.*` without cfg.upsample_triplane) plus `smplx_model.*` (buffers that
        come from the SMPL-X file).  A parameter of a module that EXISTS here and is absent from the checkpoint is an
        error -- it would silently stay at its random initialisation -- unless the checkpoint has no entry at all
        under that module's prefix and `strict` is off (a file saved without the renderer, or without the audio net).
        Returns (missing, unexpected) accumulated over both loads; with `strict` unexpected keys raise too."""
        state = path_or_state
        if isinstance(path_or_state, str):
            state = torch.load(path_or_state, map_location="cpu", weights_only=True)
        if "state_dict" in state:
            state = state["state_dict"]
        audio = {k[len("audio_triplane."):]: v for k, v in state.items() if k.startswith("audio_triplane.")}
        # the renderer appears under both prefixes (it is one shared module); take whichever is present
        rend = {k[len("triplane_gaussian.renderer."):]: v for k, v in state.items()
                if k.startswith("triplane_gaussian.renderer.")}
        drop = ["smplx_model."]  # buffers of the SMPL-X file
        if not hasattr(self.renderer, "point_encoder"):
            drop += ["point_encoder.", "point_refiner."]
        if not hasattr(self.renderer, "triplane_upsampler"):
            drop.append("triplane_upsampler.")
        drop = tuple(drop)
        # the shared renderer's tensors, from whichever prefix carries them
        for k in [k for k in audio if k.startswith("renderer.")]:
            rend.setdefault(k[len("renderer."):], audio.pop(k))
        rend = {k: v for k, v in rend.items() if not k.startswith(drop)}
        missing, unexpected = [], []
        if audio or strict:
            own = {k: v for k, v in self.audio_triplane.state_dict().items() if not k.startswith("renderer.")}
            missing += [k for k in own if k not in audio]
            unexpected += [k for k in audio if k not in own]
            self.audio_triplane.load_state_dict({k: v for k, v in audio.items() if k in own}, strict=False)
        if rend or strict:
            r = self.renderer.load_state_dict(rend, strict=False)
            missing += ["renderer." + k for k in r.missing_keys]
            unexpected += ["renderer." + k for k in r.unexpected_keys]
        if missing:
            raise KeyError(f"reference checkpoint lacks {len(missing)} tensors of modules this build runs (they would "
                           f"stay randomly initialised): {missing[:8]}{' ...' if len(missing) > 8 else ''}")
        if strict and unexpected:
            raise KeyError(f"unexpected checkpoint keys: {unexpected[:8]}")
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    # ---- one window ----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def predict_step(self, triplanes, smplx_tokens, audio_features, cam_params, ref_img_features=None):
        """lightning_model_wrapper.py:599-605,640: returns audio_rendered_images [B,T_out,H,W,3]."""
        return self.audio_triplane(audio_features, triplanes, ref_img_features, cam_params, smplx_tokens)[0]

    # ---- chained windows (demo) ------------------------------------------------------------------------------------
    @torch.no_grad()
    def rollout(self, triplanes, smplx_tokens, audio_features, cam_params, num_windows=None):
        """main2.py:179-203.  audio_features [B, W*T_out(+), 768], cam_params [B, W*T_out, ...] -> dict with
        images [B, W*T_out, H, W, 3] plus the final two token sets (to continue the clip later)."""
        T = self.audio_triplane.T_output
        W = audio_features.shape[1] // T if num_windows is None else num_windows
        if W < 1 or audio_features.shape[1] < W * T or cam_params["intrinsic"].shape[1] < W * T:
            raise ValueError(f"need {W * T} audio tokens and cameras for {W} windows of {T} frames")
        images = []
        for w in range(W):
            sl = slice(w * T, (w + 1) * T)
            cam = {k: v[:, sl] for k, v in cam_params.items()}
            out = self.audio_triplane(audio_features[:, sl], triplanes, None, cam, smplx_tokens)
            images.append(out[0])
            triplanes, smplx_tokens = out[3][:, -2:], out[4][:, -2:]  # chain into the next window
        return {"images": torch.cat(images, dim=1), "triplanes": triplanes, "smplx_tokens": smplx_tokens}

    @torch.no_grad()
    def rollout_interleaved(self, seeds, audio_features, cam_params, num_windows=None):
        """The demo's two interleaved chains (main2.py:160-311).  Every dataset item is a stride-2 clip
        (dataset_speech_vid.py:150), so the demo rolls an "even" chain over frames 0,2,4,... seeded from item 0 and an
        "odd" chain over frames 1,3,5,... seeded from item 1, and zips the two frame lists (main2.py:301-311).

        seeds = ((triplanes_even, smplx_tokens_even), (triplanes_odd, smplx_tokens_odd)), each [1,2,...];
        audio_features [1, 2*W*T_out, 768] and cam_params [1, 2*W*T_out, ...] are per output frame of the zipped clip.
        The chains are independent, so they run as ONE batch of two.  Returns images [2*W*T_out, H, W, 3]."""
        (tri_e, smpl_e), (tri_o, smpl_o) = seeds
        T = self.audio_triplane.T_output
        n = audio_features.shape[1]
        W = n // (2 * T) if num_windows is None else num_windows
        if W < 1 or n < 2 * W * T or cam_params["intrinsic"].shape[1] < 2 * W * T:
            raise ValueError(f"need {2 * W * T} audio tokens and cameras for 2 x {W} windows of {T} frames")
        unzip = lambda x: torch.cat([x[:, 0:2 * W * T:2], x[:, 1:2 * W * T:2]], dim=0)  # [2, W*T, ...]: even, odd
        out = self.rollout(torch.cat([tri_e, tri_o]), torch.cat([smpl_e, smpl_o]), unzip(audio_features),
                           {k: unzip(v) for k, v in cam_params.items()}, num_windows=W)["images"]
        return torch.stack([out[0], out[1]], dim=1).reshape(2 * W * T, *out.shape[2:])  # e0, o0, e1, o1, ...

    @torch.no_grad()
    def rollout_tokens(self, triplanes, smplx_tokens, audio_features, num_windows):
        """The token side of rollout(): yields (triplane tokens [B,T,C,3R^2], smpl tokens [B,T,D,L]) window by window,
        chained exactly like main2.py:179-203, without rendering."""
        T = self.audio_triplane.T_output
        for w in range(num_windows):
            tri, smpl = self.audio_triplane.generate_tokens(audio_features[:, w * T:(w + 1) * T], triplanes, smplx_tokens)
            yield tri, smpl
            triplanes, smplx_tokens = tri[:, -2:], smpl[:, -2:]

    @torch.no_grad()
    def rollout_sharded(self, triplanes, smplx_tokens, audio_features, cam_params, group=None, mode="segment"):
        """Multi-GPU clip: returns uint8 [world * frames_per_rank, H, W, 3] (batch item 0) on every rank.

        mode="segment" (default, SURVEY section 8e option i): every rank generates and renders a contiguous block of
        windows starting from the SAME reference tokens (as training seeds every window from the encoder,
        lightning_model_wrapper.py:435-466); only frames are exchanged.
        mode="sequential" (option ii): rank 0 runs the one exact chain of the demo loop (every window seeded by the
        previous one) and sends each rank its block of tokens as soon as it exists; ranks render their blocks while
        rank 0 keeps generating.  Same pixels as rollout() on one GPU; only the rendering scales.
        """
        import torch.distributed as dist

        world, rank = dist.get_world_size(group), dist.get_rank(group)
        T = self.audio_triplane.T_output
        windows = audio_features.shape[1] // T
        if windows % world:
            raise ValueError(f"{windows} windows do not split evenly over {world} ranks")
        w0, w1 = shard_range(windows, world, rank)
        sl = slice(w0 * T, w1 * T)
        cam = {k: v[:, sl] for k, v in cam_params.items()}
        if mode == "segment":
            local = self.rollout(triplanes, smplx_tokens, audio_features[:, sl], cam)["images"]
        elif mode == "sequential":
            if triplanes.shape[0] != 1:
                raise ValueError("sequential mode shards one clip (batch 1)")
            per = (w1 - w0) * T
            net = self.audio_triplane
            tri_shape = (per, net.cfg.triplane_feature_dim, net.triplane_token_len)
            smpl_shape = (per, net.cfg.smpl_token_dim, net.smplx_token_len)
            local = None
            if rank == 0:
                pending, block = [], []
                for w, (tri, smpl) in enumerate(self.rollout_tokens(triplanes, smplx_tokens, audio_features, windows)):
                    block.append((tri[0], smpl[0]))
                    if (w + 1) % (w1 - w0) == 0:  # a rank's block is complete: hand it over, keep generating
                        dst = w // (w1 - w0)
                        tri_b = torch.cat([b[0] for b in block]).contiguous()
                        smpl_b = torch.cat([b[1] for b in block]).contiguous()
                        if dst == 0:  # its own block comes first: enqueue its frames now, ahead of the later blocks' steps
                            local = self.renderer(tri_b.unsqueeze(0), cam, smpl_b.unsqueeze(0))[0]
                        else:
                            pending += [(send_frames(tri_b, dst, group), tri_b), (send_frames(smpl_b, dst, group), smpl_b)]
                        block = []
                for req, _keep in pending:
                    req.wait()
            else:
                dev, dt = triplanes.device, triplanes.dtype
                tri_l = recv_frames(tri_shape, dt, dev, 0, group)
                smpl_l = recv_frames(smpl_shape, dt, dev, 0, group)
                local = self.renderer(tri_l.unsqueeze(0), cam, smpl_l.unsqueeze(0))[0]
        else:
            raise ValueError(f"unknown mode {mode!r}")
        rgba = torch.cat([local[0], torch.ones_like(local[0][..., :1])], dim=-1).contiguous()
        return all_gather_frames(ops.frames_to_rgb8(rgba), group)
