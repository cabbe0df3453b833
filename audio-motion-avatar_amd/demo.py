"""Demo driver: the build's counterpart of `python -m src.main2 --mode demo` (src/main2.py:46-57,123-386).

    python -m audio_motion_avatar_amd.demo --checkpoint ckpt.pt --tokens seed_tokens.pt --audio speech.wav \
           --frames 250 --out clip.mp4

What the reference's demo does, and where it lives here:
  * load `checkpoint['state_dict']` into the two stages (main2.py:127-138)      -> harness.load_reference_checkpoint
  * stage 1 on the reference images -> triplane / SMPL-X tokens (main2.py:170-177): the stage-1 encoder is a SURVEY
    section 8(f) next-row; this driver takes its OUTPUT (`--tokens`: a file with `triplanes [1,2,C,3R^2]` and
    `smplx_tokens [1,2,D,L]`, or two such pairs for the interleaved form), or seeded random tokens
  * audio -> Wav2Vec2 features (dataset_speech_vid.py:37-116)                     -> audio_frontend (resampled to 16 kHz)
  * windows of T_out frames chained through `out[:, -2:]` (main2.py:179-203), optionally as the even / odd pair of
    stride-2 chains zipped together (main2.py:160-311)                             -> harness.rollout(_interleaved)
  * frames -> 24 fps mp4 + the clip's audio muxed in by ffmpeg (main2.py:342-384) -> FrameWriter (below)

FrameWriter needs no OpenCV: frames leave the GPU as uint8 RGB (`(frame * 255).astype(uint8)`, main2.py:351, done by
`amav_frames_to_rgb8`) and go as raw `rgb24` into a pipe.  With an `.mp4` / `.mkv` / `.mov` target the pipe is an
`ffmpeg` child process that encodes the video and muxes the audio in the same pass (the reference writes a silent mp4
with cv2 and re-muxes it); without ffmpeg on PATH that target is refused, and a `.rgb` target (raw frames + a JSON
side-car with width / height / fps) or any writable binary stream works everywhere.
"""
import argparse
import json
import shutil
import subprocess
import sys

import torch

from . import ops

VIDEO_SUFFIXES = (".mp4", ".mkv", ".mov")


class FrameWriter:
    """Sink of rendered frames: `write(frames)` takes fp32 [..., H, W, 3|4] in [0, 1] on the HIP device (or uint8 RGB
    on any device) and appends them; `close()` finishes the file (and waits for ffmpeg)."""

    def __init__(self, target, height, width, fps=24.0, audio_path=None, ffmpeg="ffmpeg"):
        self.height, self.width, self.fps = int(height), int(width), float(fps)
        self.frames = 0
        self.proc = None
        self.sidecar = None
        self._close_stream = False
        if hasattr(target, "write"):
            self.stream = target
        elif str(target).lower().endswith(VIDEO_SUFFIXES):
            exe = shutil.which(ffmpeg)
            if exe is None:
                raise RuntimeError(f"{target}: encoding needs `{ffmpeg}` on PATH (not in this image); write a .rgb file "
                                   "or pass a stream instead")
            cmd = [exe, "-y", "-f", "rawvideo", "-pix_fmt", "rgb24", "-s", f"{self.width}x{self.height}", "-r",
                   str(self.fps), "-i", "-"]
            if audio_path:  # main2.py:366-382: -i audio, -c:a aac, cut to the video's duration
                cmd += ["-i", audio_path, "-c:a", "aac", "-shortest"]
            cmd += ["-pix_fmt", "yuv420p", str(target)]
            self.proc = subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            self.stream = self.proc.stdin
        else:
            self.stream = open(target, "wb")
            self._close_stream = True
            self.sidecar = str(target) + ".json"

    def write(self, frames):
        if frames.dtype != torch.uint8:
            if frames.shape[-1] == 3:  # the reference's [B,T,H,W,3] images: add the alpha the packing kernel skips
                frames = torch.cat([frames, torch.ones_like(frames[..., :1])], dim=-1)
            frames = ops.frames_to_rgb8(frames.contiguous())
        frames = frames.reshape(-1, self.height, self.width, 3)
        self.stream.write(frames.cpu().numpy().tobytes())
        self.frames += frames.shape[0]

    def close(self):
        if self.proc is not None:
            self.stream.close()
            if self.proc.wait() != 0:
                raise RuntimeError("ffmpeg failed")
        elif self._close_stream:
            self.stream.close()
        if self.sidecar:
            with open(self.sidecar, "w") as fh:
                json.dump({"pix_fmt": "rgb24", "width": self.width, "height": self.height, "fps": self.fps,
                           "frames": self.frames}, fh)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _load_tokens(path, cfg, device, seed):
    a = cfg.triplane_audio_net
    shape_tri = (1, a.triplane_input_frames, a.triplane_feature_dim, 3 * a.triplane_resolution ** 2)
    shape_smpl = (1, a.triplane_input_frames, a.smpl_token_dim, a.smpl_token_len)
    if path is None:
        g = torch.Generator().manual_seed(seed)
        return [(torch.randn(shape_tri, generator=g).to(device), (torch.randn(shape_smpl, generator=g) * 0.1).to(device))
                for _ in range(2)]
    blob = torch.load(path, map_location="cpu", weights_only=True)
    pairs = blob if isinstance(blob, (list, tuple)) else [blob]
    out = []
    for p in pairs:
        tri, smpl = p["triplanes"].float(), p["smplx_tokens"].float()
        if tuple(tri.shape) != shape_tri or tuple(smpl.shape) != shape_smpl:
            raise ValueError(f"{path}: tokens {tuple(tri.shape)} / {tuple(smpl.shape)}, expected {shape_tri} / {shape_smpl}")
        out.append((tri.to(device), smpl.to(device)))
    return out


def main(argv=None):
    from .audio_frontend import build_wav2vec2, extract_audio_features, load_wav
    from .config import ModelConfig, RendererConfig
    from .harness import AudioDrivenAvatar
    from .synthetic import init_random_heads, make_render_inputs

    ap = argparse.ArgumentParser(description="audio-driven avatar demo on MI355X (counterpart of src.main2 --mode demo)")
    ap.add_argument("--checkpoint", help="reference checkpoint (state_dict with audio_triplane.* / triplane_gaussian.*)")
    ap.add_argument("--tokens", help="stage-1 output: torch file with `triplanes` and `smplx_tokens` (a list of two for --interleave)")
    ap.add_argument("--audio", help="PCM WAV file; without it seeded noise stands in")
    ap.add_argument("--wav2vec2", help="local wav2vec2-base-960h directory (random weights of that architecture otherwise)")
    ap.add_argument("--smplx-model-path", help="directory or file of SMPLX_NEUTRAL.npz (synthetic body otherwise)")
    ap.add_argument("--frames", type=int, default=48)
    ap.add_argument("--image-size", type=int, nargs=2, default=(512, 512), metavar=("H", "W"))
    ap.add_argument("--interleave", action="store_true", help="the demo's even / odd chains zipped (main2.py:160-311)")
    ap.add_argument("--reference-renderer", action="store_true",
                    help="the reference's default renderer.yaml: triplane upsampler x16, PTv3 point refiner, 30 000 Gaussians "
                         "(what a released checkpoint was trained with); default is BASELINE's 10 000-Gaussian renderer")
    ap.add_argument("--fps", type=float, default=24.0)
    ap.add_argument("--out", default="demo.rgb", help=".rgb (raw frames + .json), .mp4 / .mkv / .mov (needs ffmpeg) or - for stdout")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--seed", type=int, default=42)
    args = ap.parse_args(argv)

    H, W = args.image_size
    extra = dict(upsample_triplane=True, no_point_refiner=False, subdivide_steps=2) if args.reference_renderer else {}
    rcfg = RendererConfig(image_size=(H, W), device=args.device, smplx_model_path=args.smplx_model_path, **extra)
    cfg = ModelConfig(renderer=rcfg)
    torch.manual_seed(args.seed)
    model = AudioDrivenAvatar(cfg)
    if args.checkpoint:
        model.load_reference_checkpoint(args.checkpoint)
    else:
        init_random_heads(model.renderer)
    T = model.audio_triplane.T_output
    lanes = 2 if args.interleave else 1
    windows = -(-args.frames // (T * lanes))
    n = windows * T * lanes
    wav2vec = build_wav2vec2(args.wav2vec2, device=args.device, seed=args.seed)
    if args.audio:
        waveform, sr = load_wav(args.audio)
    else:
        g = torch.Generator().manual_seed(args.seed)
        waveform, sr = torch.randn(1, int(16000 * (n / 30.0 + 0.5)), generator=g) * 0.1, 16000
    audio = extract_audio_features(waveform, sr, n, wav2vec).unsqueeze(0)
    _, _, cam = make_render_inputs(n, rcfg, seed=args.seed, device=args.device)
    seeds = _load_tokens(args.tokens, cfg, args.device, args.seed)
    target = sys.stdout.buffer if args.out == "-" else args.out
    with FrameWriter(target, H, W, fps=args.fps, audio_path=args.audio) as writer:
        if args.interleave:
            if len(seeds) < 2:
                raise SystemExit("--interleave needs two token pairs (even and odd chain)")
            writer.write(model.rollout_interleaved((seeds[0], seeds[1]), audio, cam)[:args.frames])
        else:
            out = model.rollout(seeds[0][0], seeds[0][1], audio, cam)
            writer.write(out["images"][0, :args.frames])
    print(f"wrote {writer.frames} frames of {W}x{H} at {args.fps} fps to {args.out}", file=sys.stderr)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
