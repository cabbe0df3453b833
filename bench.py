#!/usr/bin/env python3
"""Benchmark of the audio-driven avatar rendering hot path on MI355X.

Workload (default, BASELINE.json configs[1]): 512x512, 10 000 Gaussians, static triplane decode + SMPL-X LBS +
rasterize for a shard of 250 frames per GPU, no audio net.  One "step" = one pass of the shard through
`Renderer.forward` (LBS -> densify/subset -> fused triplane decode -> batched tile rasterizer); all inputs are
resident in HBM before the timed region.  `value` = frames rendered by all ranks per second.

With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU) every rank renders its own 250-frame
shard (weak scaling) and the rendered sequence is reassembled on every rank by an RCCL all-gather of the uint8 RGB
frames -- by default of their non-background 16x16 tiles only (lossless, --wire sparse), unpacked to dense frames on
every rank -- issued on a side stream so that it overlaps the next step's rendering; after the timed region every rank
checks that its block of the reassembled clip equals the frames it rendered (`config.exchange_verified`).
Other workloads: --workload full (BASELINE configs[2]/[3]: the autoregressive audio net in front), --workload stress
(configs[4] per GPU).  stdout carries exactly the one JSON line (RCCL's banner and everything else go to stderr).

Extra objects on the JSON line: `roofline` (the blend kernel, timed live with HIP events around the kernel on its
own stream), `cpu_baseline` (the CPU oracle = a port, timed on this box's host cores on a bounded sample of the
same workload), `parity` (GPU vs oracle on that sample; all-pixel maxima and counts, and a `pass` flag: the process
exits non-zero after printing the line when it is false) and -- default workload at N = 1 only -- `full_path`: a few
250-frame steps of BASELINE configs[2] (the audio-driven path the >= 30 frames/s/GPU target is quoted on) run after
the timed region, with their own `roofline` (the self-attention kernel, HIP events around every launch),
`cpu_baseline` and `parity` (tokens and frames after two autoregressive steps against the oracle);
`full_path_exact_fp32`: the same clip with the exact-fp32 kernels selected per call (the split-product figure's
counterpart); `point_refiner`: the PTv3 point refiner the reference's default renderer runs on the
same frames, with the roofline of its dominant kernel, the CPU oracle's time for one frame and the parity of the
refined points; `reference_defaults`: one 6-frame window of the reference's default renderer.yaml (upsampler + refiner +
30 000 Gaussians) with parity of frame 0; `stress`: BASELINE configs[4] per-GPU shapes with the blend and projection
rooflines and parity of frame 0.  Every `parity` carries an explicit `pass`; the process exits non-zero if one is false.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


PROFILE_TAG = "r03"  # the committed rocprofv3 summaries of THIS build (tools/gpu_profiles.sh + collect_profiles.py)
# objects measured after the timed region of the default line; each may carry a `parity` with an explicit `pass`
EXTRA_OBJECTS = ("full_path", "full_path_exact_fp32", "point_refiner", "reference_defaults", "stress")


def _profile_row(filename, kernel_substr):
    """Row (dict of floats) of a kernel in a committed counter summary under profiles/, or None."""
    import csv

    path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_{filename}")
    try:
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if kernel_substr in row["kernel"]:
                    return {k: float(v) for k, v in row.items() if k not in ("kernel", "") and v not in ("", "nan")}
    except (OSError, ValueError, KeyError):
        pass
    return None


def pmc_traffic_bytes(kernel_substr):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC summary of this same command
    (profiles/r02_bench_render_pmc_hbm.csv: separate FETCH_SIZE / WRITE_SIZE passes, values in KB; see
    profiles/README.md for the gfx950 caveats).  None when the summary is absent."""
    row = _profile_row("bench_render_pmc_hbm.csv", kernel_substr)
    if not row:
        return None
    return (row["FETCH_SIZE_KB_per_launch"] + row["WRITE_SIZE_KB_per_launch"]) * 1024.0


def pmc_issue_fractions(kernel_substr):
    """Vector-ALU issue occupancy of a kernel from the committed SQ counter passes (profiles/r02_bench_render_pmc_sq.csv):
    issued VALU wave-instructions x cycles each, over (1024 SIMDs x the launch's cycles = GRBM_GUI_ACTIVE / 8 XCDs).
    A wave64 VALU instruction occupies a SIMD for 2.13 cycles with >= 5 waves resident (tools/valu_rate_probe.hip,
    profiles/r02_valu_rate_probe.txt); VERDICT r1 asked for the 4-cycle form (one wave alone), reported beside it."""
    row = _profile_row("bench_render_pmc_sq.csv", kernel_substr)
    if not row or not row.get("GRBM_GUI_ACTIVE"):
        return None
    simd_cycles = 1024.0 * row["GRBM_GUI_ACTIVE"] / 8.0
    out = {"valu_wave_instructions_per_launch": row["SQ_INSTS_VALU"], "salu_per_launch": row.get("SQ_INSTS_SALU"),
           "lds_per_launch": row.get("SQ_INSTS_LDS"), "launch_cycles": row["GRBM_GUI_ACTIVE"] / 8.0,
           "valu_issue_frac": row["SQ_INSTS_VALU"] * 2.13 / simd_cycles,
           "valu_issue_frac_4_cycle_form": row["SQ_INSTS_VALU"] * 4.0 / simd_cycles,
           "all_issue_frac": (row["SQ_INSTS_VALU"] * 2.13 + row.get("SQ_INSTS_SALU", 0.0) * 1.1 +
                              row.get("SQ_INSTS_LDS", 0.0)) / simd_cycles,
           "source": f"profiles/{PROFILE_TAG}_bench_render_pmc_sq.csv (profiled launches of the same command)"}
    return out


def pmc_mfma_busy(kernel_substr):
    """Matrix-pipe occupancy of a kernel: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch cycles), from the committed
    counter pass of tools/bench_attention.py (profiles/r02_attention_transformer_pmc_sq.csv)."""
    row = _profile_row("attention_transformer_pmc_sq.csv", kernel_substr)
    if not row or not row.get("GRBM_GUI_ACTIVE") or "SQ_VALU_MFMA_BUSY_CYCLES" not in row:
        return None
    return row["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * row["GRBM_GUI_ACTIVE"] / 8.0)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: 200 x ~1 ms for the render workload, 100 for stress, 3 x 1.2 s for full)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--frames", type=int, default=250, help="frames per GPU per step (shard size)")
    ap.add_argument("--gaussians", type=int, default=10000)
    ap.add_argument("--image", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-path", action="store_true",
                    help="default workload only: skip the `full_path` object (a few steps of configs[2] after the timed region)")
    ap.add_argument("--no-refiner", action="store_true",
                    help="default workload only: skip the `point_refiner` object (PTv3 refiner, SURVEY 8(f) row 2)")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="default workload only: skip the `reference_defaults` and `stress` objects")
    ap.add_argument("--full-steps", type=int, default=2, help="steps (250-frame clips) of the `full_path` measurement")
    ap.add_argument("--cpu-frames", type=int, default=8, help="frames of the workload timed on the CPU oracle")
    ap.add_argument("--chunks", type=int, default=1,
                    help="frame groups pipelined on separate HIP streams (1 is fastest; 2 costs 17 percent more time, 4 costs 56 percent)")
    ap.add_argument("--exchange", choices=["collective", "direct"], default=os.environ.get("AMAV_EXCHANGE", "collective"),
                    help="frame all-gather as RCCL's collective or as grouped peer-to-peer sends (mesh form)")
    ap.add_argument("--wire", choices=["sparse", "dense"], default="sparse",
                    help="N > 1 exchange format: non-background 16x16 tiles of the uint8 frames (lossless) or all of them")
    ap.add_argument("--workload", choices=["render", "full", "stress"], default="render",
                    help="render = BASELINE configs[1] (static decode + LBS + rasterize, the metric's config); "
                         "full = configs[2]: synthetic audio tokens -> AudioTriplaneNet (autoregressive) -> SMPL-X "
                         "decoder -> LBS -> decode -> rasterize; stress = configs[4] per GPU: triplane 128^2 x 512 "
                         "channels, 50 000 Gaussians, 1024x1024, decode + LBS + rasterize (32 frames per step)")
    return ap.parse_args()


def renderer_config(args, device, with_decoder=False):
    from audio_motion_avatar_amd.config import RendererConfig

    stress = args.workload == "stress"
    steps = {10000: 0, 30000: 1, 50000: 2}.get(args.gaussians)
    if steps is None or (args.gaussians == 50000 and not stress):
        raise SystemExit("--gaussians must be 10000 or 30000 (the reference's SUBDEVIDE_VERTS table)")
    cfg = RendererConfig(image_size=(args.image, args.image), subdivide_steps=steps,
                         predict_smplx_params=with_decoder, device=device)
    if stress:  # BASELINE configs[4]
        cfg.triplane_resolution, cfg.triplane_feature_dim, cfg.num_gaussians = 128, 512, args.gaussians
    return cfg


def build_renderer(args, device, with_decoder=False):
    from audio_motion_avatar_amd.renderer import Renderer
    from audio_motion_avatar_amd.smplx_decoder import SMPLXDecoder
    from audio_motion_avatar_amd.synthetic import init_random_heads

    cfg = renderer_config(args, device, with_decoder)
    dec = SMPLXDecoder(cfg).to(device) if with_decoder else None
    return init_random_heads(Renderer(cfg, smpl_decoder=dec).eval()), cfg


# fp32 tensors end to end; the transformer's matrix products are fp32-equivalent sums of fp16 partial products with
# fp32 accumulation, everything else plain fp32
TRANSFORMER_DTYPE = "f32 (transformer products: fp16 x 2 split operands, fp32 accumulate)"
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA = the fp32 vector peak
MFMA_16BIT_PEAK_TFLOPS = 2516.6  # dense bf16 / fp16 MFMA: 32x32x16 in 32 cycles per SIMD, 1024 SIMDs, 2.4 GHz


class FullPath:
    """BASELINE configs[2]: 16 kHz waveform -> Wav2Vec2 (base-960h architecture, random weights: no checkpoint is
    available offline) -> AudioTriplaneNet rolled with the demo's semantics (windows of T_out = 6 frames, every
    window seeded with the previous one's last two outputs, src/main2.py:179-203) -> SMPLXDecoder -> SMPL-X LBS ->
    fused decode -> tile rasterizer, `frames` frames per step in one batched render."""

    def __init__(self, args, device, rank, frames):
        from audio_motion_avatar_amd.audio_frontend import build_wav2vec2
        from audio_motion_avatar_amd.config import ModelConfig
        from audio_motion_avatar_amd.harness import AudioDrivenAvatar
        from audio_motion_avatar_amd.synthetic import init_random_heads, make_render_inputs

        rcfg = renderer_config(args, device, with_decoder=True)
        torch.manual_seed(1234)  # random-init weights of the reference architecture (no checkpoint exists here)
        self.avatar = AudioDrivenAvatar(ModelConfig(renderer=rcfg))
        self.renderer, self.net, self.rcfg = init_random_heads(self.avatar.renderer), self.avatar.audio_triplane, rcfg
        with torch.no_grad():  # keep the autoregressive chain bounded, as a trained model's would be: each step then
            self.net.transformer.proj_out.weight.mul_(0.02)  # perturbs the N(0,1) tokens instead of compounding them
            self.net.transformer.proj_out.bias.zero_()
            self.renderer.smpl_decoder.dec_transl.bias.copy_(torch.tensor([0.0, -0.15, 2.4]))  # body in front of the camera
            dec = self.renderer.smpl_decoder
            dec.dec_body_root_pose.bias.copy_(torch.tensor([1.0, 0, 0, 0, -1.0, 0]))  # upright for the y-down camera
            for head in (dec.dec_body_pose, dec.dec_hand_pose, dec.dec_face_jaw_pose, dec.dec_leye_pose, dec.dec_reye_pose):
                head.weight.mul_(0.2)  # joint rotations near the identity (6-D rows (1,0,0),(0,1,0)) + a perturbation
                head.bias.copy_(torch.tensor([1.0, 0, 0, 0, 1.0, 0]).repeat(head.bias.numel() // 6))
        self.F, self.device = frames, device
        self.T = self.net.T_output
        self.windows = (frames + self.T - 1) // self.T
        g = torch.Generator().manual_seed(42 + rank)
        # the front-end crops the waveform to the clip at its hard-coded 30 fps (dataset_speech_vid.py:54)
        self.wav2vec = build_wav2vec2(device=device, seed=7)
        self.waveform = (torch.randn(1, int(16000 * (self.windows * self.T / 30.0 + 0.5)), generator=g) * 0.1).to(device)
        self.tri = torch.randn(1, 2, 256, 3 * 32 * 32, generator=g).to(device)
        self.smpl_tok = (torch.randn(1, 2, 256, 80, generator=g) * 0.1).to(device)
        _, _, self.cam = make_render_inputs(frames, rcfg, seed=42 + rank, device=device)
        self.workspaces = [None] * max(1, min(args.chunks, frames))
        self.chunks = args.chunks

    def tokens(self):
        from audio_motion_avatar_amd.audio_frontend import extract_audio_features

        n = self.windows * self.T
        audio = extract_audio_features(self.waveform, 16000, n, self.wav2vec).unsqueeze(0)  # [1, n, 768]
        tri, smpl = zip(*self.avatar.rollout_tokens(self.tri, self.smpl_tok, audio, self.windows))
        return torch.cat(tri, dim=1)[:, :self.F], torch.cat(smpl, dim=1)[:, :self.F]

    def render(self, out_tri, out_smpl):
        F = out_tri.shape[1]
        params = self.renderer.smpl_decoder(out_smpl.reshape(F, 256, 80))
        params = {k: v.reshape(1, F, *v.shape[1:]) for k, v in params.items()}
        cam = {k: v[:, :F] for k, v in self.cam.items()}
        ws = self.workspaces if F == self.F else [None]
        return self.renderer.render_tokens(out_tri[0], params, cam, chunks=self.chunks if F == self.F else 1,
                                           workspaces=ws, check_overflow=F != self.F)[0]

    def step(self):
        with torch.no_grad():
            return self.render(*self.tokens())

    def size_workspaces(self):
        from audio_motion_avatar_amd import ops

        self.step()
        for ci, ws in enumerate(self.workspaces):
            _, max_frame, over = ws.status_full()
            if over:
                fc, n, h, w = ws.key
                self.workspaces[ci] = ops.RasterWorkspace(fc, n, h, w, int(fc * max_frame * 1.25), self.device)

    def attention_roofline(self, ar_steps=3):
        """HIP events around every self-attention launch (operand split + flash kernel + split-key combine) of
        `ar_steps` transformer passes at the full shape."""
        from audio_motion_avatar_amd import ops

        net, L = self.net, len(self.net.transformer.transformer_blocks)
        S, H = 2 * (self.net.triplane_token_len + self.net.smplx_token_len), self.net.cfg.transformer_num_heads
        q = torch.randn(1, 256, S, device=self.device)
        a = torch.randn(1, 1, 768, device=self.device)
        events = [(ops.Event(), ops.Event()) for _ in range(L * ar_steps)]
        with torch.no_grad():
            net.transformer(q, a)  # warm-up
            ops.ATTN_PROFILE_EVENTS = list(events)
            t0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0[0].record()
            for _ in range(ar_steps):
                net.transformer(q, a)
            t0[1].record()
            ops.ATTN_PROFILE_EVENTS = None
        torch.cuda.synchronize()
        ms = sum(s.elapsed_ms(e) for s, e in events) / len(events)
        flop = 4.0 * S * S * 64 * H
        step_ms = t0[0].elapsed_time(t0[1]) / ar_steps
        # one AR step: L x (q/k/v + out projections 4 * 2 S 512^2, GEGLU feed-forward 2 S 512 (4096 + 2048), attention)
        # + proj_in / proj_out 2 * 2 S 256 512
        step_flop = L * (flop + 2.0 * S * 512 * (4 * 512 + 4096 + 2048)) + 4.0 * S * 256 * 512
        variant = os.environ.get("AMAV_ATTN", "fp16")
        tf = flop / (ms * 1e-3) / 1e12
        step_tf = step_flop / (step_ms * 1e-3) / 1e12
        out = {"bound": "mfma", "unit": "TFLOP/s", "traffic": None, "avg_launch_ms": ms,
               "algorithmic_flop_per_launch": flop, "launches_timed": len(events)}
        if variant == "f32":
            out.update({"kernel": "selfattn_kernel + combine_kernel (fp32 MFMA flash attention, S=%d, H=%d)" % (S, H),
                        "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "frac": tf / MFMA_F32_PEAK_TFLOPS,
                        "mfma_busy_frac": pmc_mfma_busy("selfattn_kernel"),
                        "transformer_step": {"ms": step_ms, "flop": step_flop, "achieved": step_tf,
                                             "frac": step_tf / MFMA_F32_PEAK_TFLOPS}})
            return out
        # fp32 in, fp32-equivalent out (1.1e-7 max abs against fp64 at the reference shape, tools/attention_accuracy.py;
        # the library's fp32 SDPA: 4.8e-7), computed as `products` low-precision partial products per fp32 product on the
        # 16-bit matrix pipe.  `achieved` / `frac` price the FLOP actually ISSUED against THAT pipe's dense peak -- the
        # figure that says how well the kernel uses the hardware it runs on; `fp32_equivalent` is the algorithmic rate,
        # next to the fp32 MFMA peak it no longer is bounded by.
        products, kname, parts = (6, "selfattn_split_kernel", "bf16 x 3") if variant == "bf16" else (3, "selfattn_f16_kernel", "fp16 x 2")
        out.update({"kernel": "operand split + %s + combine_kernel (flash attention on %s split operands, %d partial "
                              "products per fp32 product, fp32-equivalent result, S=%d, H=%d)" % (kname, parts, products, S, H),
                    "issued_flop_per_launch": products * flop, "achieved": products * tf, "peak": MFMA_16BIT_PEAK_TFLOPS,
                    "frac": products * tf / MFMA_16BIT_PEAK_TFLOPS, "mfma_busy_frac": pmc_mfma_busy(kname),
                    "fp32_equivalent": {"achieved": tf, "fp32_mfma_peak": MFMA_F32_PEAK_TFLOPS,
                                        "ratio_to_fp32_mfma_peak": tf / MFMA_F32_PEAK_TFLOPS},
                    "transformer_step": {"ms": step_ms, "flop": step_flop, "achieved_fp32_equivalent": step_tf,
                                         "ratio_to_fp32_mfma_peak": step_tf / MFMA_F32_PEAK_TFLOPS,
                                         "note": "projections as fp16 x 2 split GEMMs (3 partial products), attention as "
                                                 "above; proj_in / proj_out and the small cross-attention rows in fp32"}})
        return out

    def cpu_baseline_and_parity(self, ar_steps=2):
        """Two autoregressive steps + their two rendered frames on the CPU oracle (a port), timed on this box's host
        cores, and the HIP path's tokens / frames against them on identical inputs and weights.  The oracle's outputs
        are kept, so a second call (the exact-fp32 kernels) compares against the same reference without re-running it."""
        from audio_motion_avatar_amd.audio_frontend import extract_audio_features
        from oracle import smplx_decoder as o_dec, transformer as o_tr

        cores = host_cores()
        torch.set_num_threads(cores)
        with torch.no_grad():
            audio = extract_audio_features(self.waveform, 16000, self.windows * self.T, self.wav2vec).unsqueeze(0)
            got_tri, got_smpl = self.net.generate_tokens(audio, self.tri, self.smpl_tok, num_steps=ar_steps)
            got_rgba = self.render(got_tri, got_smpl)
            if getattr(self, "_oracle", None) is not None and self._oracle[0] == ar_steps:
                _, ref_tri, ref_smpl, img, unstable, base = self._oracle
                return base, self._parity(ar_steps, got_tri, got_smpl, got_rgba, ref_tri, ref_smpl, img, unstable)
            p = {k: v.detach().cpu() for k, v in self.net.state_dict().items() if not k.startswith("renderer.")}
            t0 = time.perf_counter()
            ref_tri, ref_smpl = o_tr.audio_triplane_tokens(p, audio.cpu(), self.tri.cpu(), self.smpl_tok.cpu(),
                                                           t_output=ar_steps)
            ar_sec = (time.perf_counter() - t0) / ar_steps
            dp = {"smpl_decoder." + k: v.detach().cpu() for k, v in self.renderer.smpl_decoder.state_dict().items()}
            ref_params = o_dec.smplx_decoder_forward(dp, ref_smpl[0])
            ref_params = {k: v.reshape(1, ar_steps, *v.shape[1:]) for k, v in ref_params.items()}
            cam = {k: v[:, :ar_steps].cpu() for k, v in self.cam.items()}
            t0 = time.perf_counter()
            img, alpha, unstable = oracle_render(self.renderer, self.rcfg, ref_tri, ref_params, cam)
            render_sec = (time.perf_counter() - t0) / ar_steps
        base = {"value": 1.0 / (ar_sec + render_sec), "unit": "frames/s", "cores": cores, "kind": "port",
                "sample": f"{ar_steps} autoregressive steps of the full-size net (8 layers, S=6304) + their {ar_steps} "
                          f"rendered 512x512 frames through oracle/ (torch CPU, {cores} threads; C rasterizer with OpenMP): "
                          f"{ar_sec:.2f} s per transformer step + {render_sec:.2f} s per frame for decode/LBS/raster"}
        self._oracle = (ar_steps, ref_tri, ref_smpl, img, unstable, base)
        return base, self._parity(ar_steps, got_tri, got_smpl, got_rgba, ref_tri, ref_smpl, img, unstable)

    @staticmethod
    def _parity(ar_steps, got_tri, got_smpl, got_rgba, ref_tri, ref_smpl, img, unstable):
        scale = float(ref_tri.abs().max())
        d_rgb = (got_rgba[..., :3].cpu() - img[0]).abs()
        parity = {"ar_steps": ar_steps, "token_scale": scale,
                  "triplane_tokens_max_abs": float((got_tri.cpu() - ref_tri).abs().max()),
                  "smpl_tokens_max_abs": float((got_smpl.cpu() - ref_smpl).abs().max()),
                  "token_tolerance": 2e-5 * max(1.0, scale),
                  "rgb": pixel_report(d_rgb.amax(-1), unstable[0]), "raster_tolerance": 1e-3}
        # tokens within the stated tolerance; no unflagged pixel above 1e-3 and no more pixels above it than the oracle
        # flagged as sitting on a blend threshold (the rule of the render workload's `parity`)
        parity["pass"] = bool(max(parity["triplane_tokens_max_abs"], parity["smpl_tokens_max_abs"]) <= parity["token_tolerance"]
                              and parity["rgb"]["unflagged_pixels_above_tolerance"] == 0
                              and parity["rgb"]["pixels_above_tolerance"] <= parity["rgb"]["pixels_flagged"]
                              and parity["rgb"]["max_abs_all_pixels"] <= 1.2e-2)
        return parity


def pixel_report(diff, unstable, tol=1e-3):
    """All-pixel statistics of a rasterizer comparison.  `unstable` = pixels the oracle flags as sitting within 1e-4
    (relative) of one of the algorithm's discontinuities (alpha < 1/255, T' < 1e-4, power > 0), where two correct
    fp32 evaluations may branch differently."""
    above = diff > tol
    return {"max_abs_all_pixels": float(diff.max()), "max_abs_unflagged_pixels": float((diff * ~unstable).max()),
            "pixels": int(diff.numel()), "pixels_above_tolerance": int(above.sum()),
            "unflagged_pixels_above_tolerance": int((above & ~unstable).sum()), "pixels_flagged": int(unstable.sum())}


def oracle_render(renderer, cfg, tokens, smpl_params, cam):
    """tokens [1,T,C,3R^2] + SMPL-X params [1,T,...] + cameras (all CPU) through oracle/: LBS -> densify -> sample +
    heads -> rasterize.  -> (rgb [1,T,H,W,3], alpha [1,T,H,W], unstable [1,T,H,W])."""
    from oracle import lbs as o_lbs, rasterizer as o_rast, subdivide as o_sub, triplane as o_tri

    model = renderer.smplx_model.oracle_arrays(torch.float32)
    levels = o_sub.subdivision_levels(renderer.smplx_model.faces, renderer.smplx_model.num_verts,
                                      max(1, cfg.subdivide_steps))
    params = {"gaussian_decoder." + k: v.detach().cpu() for k, v in renderer.gaussian_decoder.state_dict().items()}
    pts = o_lbs.get_smpl_vertices(model, smpl_params, densify=(levels, renderer.subset_index))
    planes = o_tri.tokens_to_planes(tokens, cfg.triplane_resolution)
    g = o_tri.decode_gaussians(params, planes, pts, smpl_params["transl"].reshape(-1, 3), cfg.radius)
    return o_rast.render_batch(g, cam["intrinsic"], cam["extrinsic"], cfg.image_size, full=True)


def measure_point_refiner(args, device, with_cpu, frames=32):
    """The `point_refiner` object of the default line (the reference's default renderer runs it,
    renderer.py:143-151): the same synthetic frames with cfg.no_point_refiner=False (ptv3_encoder.yaml), LBS points ->
    triplane features -> PointTransformerV3 -> offset MLP, timed with HIP events; roofline of its dominant kernel (the
    stem's 5x5x5 gather-GEMM on fp32 MFMA), CPU oracle on one frame beside it."""
    import dataclasses

    from audio_motion_avatar_amd import ops
    from audio_motion_avatar_amd.point_transformer import Level
    from audio_motion_avatar_amd.renderer import Renderer
    from audio_motion_avatar_amd.synthetic import init_random_heads, make_render_inputs

    cfg = dataclasses.replace(renderer_config(args, device), no_point_refiner=False)
    torch.manual_seed(7)
    r = init_random_heads(Renderer(cfg).eval())
    with torch.no_grad():
        r.point_refiner[-1].weight.normal_(0, 0.01)  # the reference zero-initialises it (offsets == 0 until trained)
        tokens, smpl, _ = make_render_inputs(frames, cfg, seed=42, device=device)
        verts = ops.points_gather(r._posed_vertices(smpl), r._gather_idx)
        r.refine_points(tokens[0], verts)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        reps = 3
        for _ in range(reps):
            refined = r.refine_points(tokens[0], verts)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / reps
        # the dominant kernel alone: the stem convolution of one pass (clouds_per_pass frames)
        per = min(frames, cfg.refiner_clouds_per_pass)
        pts = verts[:per].contiguous()
        n = per * pts.shape[1]
        cloud_of = torch.arange(per, device=device, dtype=torch.int32).repeat_interleave(pts.shape[1])
        grid, depth = ops.cloud_voxelize(pts.reshape(n, 3), cloud_of, per)
        import numpy as np
        level = Level(grid, cloud_of, depth, np.full(per, pts.shape[1]), ops.cloud_codes(grid, cloud_of, depth))
        stem = r.point_encoder.point_transformer.embedding.stem.conv
        pairs = level.pairs(5)
        feat = torch.randn(n, stem.in_channels, device=device)
        w = stem.tap_weights()
        split = os.environ.get("AMAV_SUBM", "split") != "f32"  # what SubMConv3d.forward runs
        ws = ops.subm_prepare_weights_split(w) if split else None

        def gemm():
            if split:
                return ops.subm_pair_gemm_split(feat, pairs.pair_src, pairs.tap_start, pairs.tile_start, pairs.tiles, ws,
                                                w.shape[0], w.shape[2])
            return ops.subm_pair_gemm(feat, pairs.pair_src, pairs.tap_start, pairs.tile_start, pairs.tiles, w)

        for _ in range(2):
            gemm()
        ev[0].record()
        for _ in range(10):
            gemm()
        ev[1].record()
        torch.cuda.synchronize()
        gemm_ms = ev[0].elapsed_time(ev[1]) / 10
    flops = 2.0 * pairs.count * stem.in_channels * stem.out_channels
    tf = flops / (gemm_ms * 1e-3) / 1e12
    N = verts.shape[1]
    common = {"bound": "mfma", "unit": "TFLOP/s", "avg_launch_ms": gemm_ms, "algorithmic_flop_per_launch": flops,
              "voxel_pairs": pairs.count, "taps_hit_of_125": pairs.count / n, "traffic": None}
    if split:  # three fp16 partial products per fp32 product: priced as issued work against the 16-bit pipe (as full_path)
        roof = dict(common, kernel="feature absmax + pair_gemm_f16_kernel<32> (stem 5x5x5 submanifold convolution, 768 -> 32, "
                                   "fp16 x 2 split products, fp32-equivalent result)",
                    issued_flop_per_launch=3.0 * flops, achieved=3.0 * tf, peak=MFMA_16BIT_PEAK_TFLOPS,
                    frac=3.0 * tf / MFMA_16BIT_PEAK_TFLOPS,
                    fp32_equivalent={"achieved": tf, "fp32_mfma_peak": MFMA_F32_PEAK_TFLOPS,
                                     "ratio_to_fp32_mfma_peak": tf / MFMA_F32_PEAK_TFLOPS})
    else:
        roof = dict(common, kernel="pair_gemm_kernel<32> (stem 5x5x5 submanifold convolution, 768 -> 32)", achieved=tf,
                    peak=MFMA_F32_PEAK_TFLOPS, frac=tf / MFMA_F32_PEAK_TFLOPS)
    out = {"workload": f"reference ptv3_encoder.yaml (5 stages, 46 M parameters, patch 512) on {N} points per frame, "
                       f"{frames} frames in passes of {per}; random weights, refiner output layer N(0, 0.01)",
           "ms_per_frame": ms / frames, "frames_per_s": frames / (ms * 1e-3), "roofline": roof}
    if with_cpu:
        from oracle import lbs as o_lbs, subdivide as o_sub, triplane as o_tri

        params = {k: v.detach().cpu() for k, v in r.state_dict().items()}
        planes = o_tri.tokens_to_planes(tokens[:, :1].cpu(), cfg.triplane_resolution)
        pcfg = {k: list(getattr(cfg, k)) for k in ("enc_depths", "enc_num_head", "enc_patch_size", "dec_depths",
                                                   "dec_num_head", "dec_patch_size")}
        t0 = time.perf_counter()
        want = o_tri.refine_points(params, planes, verts[:1].cpu(), cfg.radius, pcfg)
        cpu_s = time.perf_counter() - t0
        err = float((refined[:1].cpu() - want).abs().max())
        moved = float((want - verts[:1].cpu()).abs().max())
        out["cpu_baseline"] = {"value": 1.0 / cpu_s, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "oracle/ptv3.py + oracle/triplane.py refine_points on frame 0 (torch CPU fp32)"}
        out["parity"] = {"refined_points_max_abs_diff_m": err, "tol": 1e-5, "largest_offset_m": moved,
                         "pass": bool(err <= 1e-5)}
    return out


def measure_full_path(args, device, rank, steps, warmup, with_cpu):
    """The `full_path` object of the default line: a few steps of configs[2] after the render measurement."""
    fp = FullPath(args, device, rank, args.frames)
    fp.size_workspaces()
    for _ in range(warmup):
        fp.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        rgba = fp.step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert not any(ws.status()[1] for ws in fp.workspaces), "rasterizer workspace overflowed"
    out = {"workload": "BASELINE configs[2]: synthetic 16 kHz audio -> Wav2Vec2 (random weights) -> AudioTriplaneNet "
                       f"(8 layers, S=6304, {fp.windows} chained windows of {fp.T} autoregressive steps, demo semantics) "
                       f"-> SMPLXDecoder -> LBS -> decode -> rasterize {fp.F} x 512x512, all inside the step",
           "metric": "rendered frames/sec, audio-driven path", "value": fp.F * steps / elapsed, "unit": "frames/s",
           "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "frames_per_step": fp.F,
           "ms_per_frame": elapsed / steps / fp.F * 1e3, "target_frames_per_s_per_gpu": 30.0,
           "dtype": TRANSFORMER_DTYPE, "output_finite": bool(torch.isfinite(rgba).all()),
           "coverage": float((rgba[..., 3] > 0.5).float().mean()),
           "weights": "random init (transformer.proj_out scaled by 0.02 so the AR chain stays bounded)",
           "roofline": fp.attention_roofline()}
    if with_cpu:
        out["cpu_baseline"], out["parity"] = fp.cpu_baseline_and_parity()
    return out, fp


class exact_fp32:
    """Context: every selectable product of the path in exact fp32 -- attention on the fp32 MFMA kernel, the LBS blend
    product on the fp32 MFMA kernel (amav_set_option, per call) and the projections as library fp32 GEMMs (AMAV_GEMM, read
    per call) -- instead of the default fp16 x 2 split products."""

    def __enter__(self):
        from audio_motion_avatar_amd import ops

        self.saved = {k: os.environ.get(k) for k in ("AMAV_GEMM", "AMAV_SUBM")}
        os.environ["AMAV_GEMM"] = os.environ["AMAV_SUBM"] = "f32"
        ops.set_option("attn", "f32")
        ops.set_option("lbs", "f32")

    def __exit__(self, *exc):
        from audio_motion_avatar_amd import ops

        ops.set_option("attn", "default")
        ops.set_option("lbs", "default")
        for k, v in self.saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def measure_full_path_exact_fp32(fp, split_line, steps, with_cpu):
    """The same clip, weights and process as `full_path` with the exact-fp32 kernels selected per call: the figure that
    stands beside the split-product one (the headline's precision must be visible on the line)."""
    with exact_fp32():
        fp.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            rgba = fp.step()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        out = {"workload": split_line["workload"], "value": fp.F * steps / elapsed, "unit": "frames/s", "steps": steps,
               "ms_per_step": elapsed / steps * 1e3, "ms_per_frame": elapsed / steps / fp.F * 1e3,
               "dtype": "f32 (exact products: attention and LBS blend on v_mfma_f32_32x32x2_f32, projections as library fp32 GEMMs)",
               "selected_by": "amav_set_option(attn=f32, lbs=f32) + AMAV_GEMM=f32, per call in this process",
               "output_finite": bool(torch.isfinite(rgba).all()),
               "split_product_speedup": split_line["value"] / (fp.F * steps / elapsed)}
        if with_cpu:
            _, out["parity"] = fp.cpu_baseline_and_parity()
    return out


def measure_reference_defaults(args, device, with_cpu, frames=6):
    """The `reference_defaults` object: one window of the reference's DEFAULT renderer configuration
    (/root/reference/src/configs/model/renderer.yaml:10-17 -- upsample_triplane: true (4 blocks, 32^2 -> 512^2), point
    refiner on, subdivide_steps: 2 -> 30 000 Gaussians) of `frames` = T_output frames (triplane_audio_net.py:269) at
    512 x 512 through Renderer.forward; parity of the refined points and of the frame against the CPU oracle on frame 0
    (full-plane upsampler + PTv3 + decode + rasterizer: the windowed upsampler must agree where the points sample)."""
    from audio_motion_avatar_amd.config import RendererConfig
    from audio_motion_avatar_amd.renderer import Renderer
    from audio_motion_avatar_amd.synthetic import init_random_heads, make_render_inputs

    cfg = RendererConfig(image_size=(args.image, args.image), subdivide_steps=2, upsample_triplane=True,
                         no_point_refiner=False, predict_smplx_params=False, device=device)
    torch.manual_seed(11)
    r = init_random_heads(Renderer(cfg).eval())
    with torch.no_grad():
        r.point_refiner[-1].weight.normal_(0, 0.005)  # the reference zero-initialises it (offsets == 0 until trained)
        tokens, smpl, cam = make_render_inputs(frames, cfg, seed=42, device=device)
        dummy = torch.zeros(1, frames, 1, 1, device=device)
        for _ in range(2):
            images, gauss = r(tokens, cam, dummy, smpl)
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            images, gauss = r(tokens, cam, dummy, smpl)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
    plan = getattr(r, "last_window_plan", None)
    out = {"workload": f"reference renderer.yaml defaults: TriplaneUpsampler x16 (C=256, 32^2 -> 512^2) + PTv3 point refiner "
                       f"+ {r.num_verts} Gaussians, {frames} frames (one T_output window) at {args.image}x{args.image} through "
                       "Renderer.forward; random weights, refiner output layer N(0, 0.005)",
           "ms_per_window": ms, "ms_per_frame": ms / frames, "frames_per_s": frames / (ms * 1e-3),
           "with_transformer_step_frames_per_s": None, "output_finite": bool(torch.isfinite(images).all()),
           "coverage": float((images < 0.999).any(-1).float().mean()),
           "upsampler_active_tiles_of_64_per_plane": [round(int(w["mask"].sum()) / frames, 1) for w in plan] if plan else None}
    if with_cpu:
        from oracle import lbs as o_lbs, rasterizer as o_rast, subdivide as o_sub, triplane as o_tri

        cores = host_cores()
        torch.set_num_threads(cores)
        params = {k: v.detach().cpu() for k, v in r.state_dict().items()}
        sp = {k: v[:, :1].cpu() for k, v in smpl.items()}
        levels = o_sub.subdivision_levels(r.smplx_model.faces, r.smplx_model.num_verts, max(1, cfg.subdivide_steps))
        pcfg = {k: list(getattr(cfg, k)) for k in ("enc_depths", "enc_num_head", "enc_patch_size", "dec_depths",
                                                   "dec_num_head", "dec_patch_size")}
        t0 = time.perf_counter()
        with torch.no_grad():
            cpu_pts = o_lbs.get_smpl_vertices(r.smplx_model.oracle_arrays(torch.float32), sp, densify=(levels, r.subset_index))
            # Stage-by-stage on identical inputs, as in the render workload's `parity`: the refiner voxelises its points
            # (floor(100 p), point_encoder.py:33), so two LBS results 1e-6 m apart put a handful of the 30 000 points
            # into different voxels and the network's output moves by centimetres around them
            # (tools/debug_reference_defaults.py).  The LBS stage has its own bar; the stages behind it get the HIP
            # path's points.
            pts = r.get_smpl_vertices({k: v[:, :1] for k, v in smpl.items()}).cpu()
            lbs_err = float((pts - cpu_pts).abs().max())
            planes = o_tri.tokens_to_planes(tokens[:, :1].cpu(), cfg.triplane_resolution)
            up = o_tri.triplane_upsampler(params, planes, cfg.num_upsample_blocks)
            refined = o_tri.refine_points(params, up, pts, cfg.radius, pcfg)
            g = o_tri.decode_gaussians(params, up, refined, sp["transl"].reshape(-1, 3), cfg.radius)
            img, alpha, unstable = o_rast.render_batch(g, cam["intrinsic"][:, :1].cpu(), cam["extrinsic"][:, :1].cpu(),
                                                       cfg.image_size, full=True)
        cpu_s = time.perf_counter() - t0
        got_xyz = gauss["xyz"][:1].cpu()
        want_xyz = g["xyz"]
        d_pts = float((got_xyz - want_xyz).abs().max())
        d_attr = max(float((gauss[k][:1].cpu() - g[k]).abs().max()) for k in ("scale", "rot", "opacity", "color"))
        rgb = pixel_report((images[0, :1].cpu() - img[0]).abs().amax(-1), unstable[0])
        out["cpu_baseline"] = {"value": 1.0 / cpu_s, "unit": "frames/s", "cores": cores, "kind": "port",
                               "sample": "frame 0 through oracle/: LBS + densify, full-plane upsampler (torch CPU convolutions), "
                                         "PTv3 refiner, decode, C rasterizer"}
        # the upsampled planes come from library convolutions on both sides (other kernels, other summation order: 5e-6),
        # sampled at 512^2 and fed to 256-channel heads: measured 2e-6 m on the Gaussians' positions, 5e-5 on the others
        out["parity"] = {"lbs_points_max_abs": lbs_err, "lbs_tolerance": 1e-5, "gaussian_xyz_max_abs_m": d_pts,
                         "xyz_tolerance": 1e-5, "other_attributes_max_abs": d_attr, "attribute_tolerance": 2e-4,
                         "largest_refiner_offset_m": float((refined - pts).abs().max()), "rgb": rgb, "raster_tolerance": 1e-3}
        out["parity"]["pass"] = bool(lbs_err <= 1e-5 and d_pts <= 1e-5 and d_attr <= 2e-4 and
                                     rgb["unflagged_pixels_above_tolerance"] == 0 and
                                     rgb["pixels_above_tolerance"] <= rgb["pixels_flagged"] and
                                     rgb["max_abs_all_pixels"] <= 1.2e-2)
    return out


def measure_stress(args, device, with_cpu, frames=32):
    """The `stress` object: BASELINE configs[4] per-GPU shapes (triplane 128^2 x 512 channels, 50 000 Gaussians, 1024 x 1024,
    static decode + LBS + rasterize) for `frames` frames per step, with the rooflines of the two kernels that dominate it
    (slab projection: HBM stream; blend: HBM by bytes) and the parity of frame 0 against the CPU oracle."""
    import dataclasses

    from audio_motion_avatar_amd import ops
    from audio_motion_avatar_amd.synthetic import make_render_inputs

    sargs = argparse.Namespace(**{**vars(args), "workload": "stress", "gaussians": 50000, "image": 1024})
    renderer, cfg = build_renderer(sargs, device)
    F, N, H, W = frames, 50000, 1024, 1024
    tokens, smpl, cam = make_render_inputs(F, cfg, seed=42, device=device)
    ws = [None]
    ev = [(ops.Event(), ops.Event())]
    with torch.no_grad():
        rgba, packed = renderer.render_tokens(tokens[0], smpl, cam, workspaces=ws)  # sizes the workspace
        steps = 10
        marks = [ops.Event() for _ in range(steps + 1)]
        blend_ms = []
        torch.cuda.synchronize()
        marks[0].record()
        for i in range(steps):
            ops.PROFILE_EVENTS = list(ev)
            rgba, packed = renderer.render_tokens(tokens[0], smpl, cam, workspaces=ws, check_overflow=False)
            marks[i + 1].record()
            torch.cuda.synchronize()
            blend_ms.append(ev[0][0].elapsed_ms(ev[0][1]))
        ops.PROFILE_EVENTS = None
        assert not ws[0].status()[1], "rasterizer workspace overflowed"
        step_ms = sorted(marks[i].elapsed_ms(marks[i + 1]) for i in range(steps))[steps // 2]
        proj_roofline = project_roofline(renderer, cfg, tokens[0], smpl, F)
    blend = sorted(blend_ms)[len(blend_ms) // 2]
    blend_bytes = F * (16 * H * W + 40 * N)
    out = {"workload": f"BASELINE configs[4] per GPU: triplane 128^2 x 512 ch, 50k Gaussians, 1024x1024, {F} frames per step, "
                       "static triplane decode + LBS + rasterize, no audio net",
           "frames_per_s": F / (step_ms * 1e-3), "ms_per_step": step_ms, "ms_per_frame": step_ms / F,
           "instances_per_step": int(ws[0].status()[0]),
           "roofline_blend": {"bound": "hbm", "kernel": "render_kernel (tile blend)", "achieved": blend_bytes / (blend * 1e-3) / 1e9,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": blend_bytes / (blend * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "avg_launch_ms": blend, "algorithmic_bytes_per_launch": blend_bytes, "traffic": None},
           "roofline_project": proj_roofline}
    if with_cpu:
        base, parity = cpu_baseline_and_parity(renderer, cfg, tokens, smpl, cam, [packed, rgba], 1)
        out["cpu_baseline"], out["parity"] = base, parity
    del tokens, rgba, packed
    return out


def project_roofline(renderer, cfg, tokens0, smpl, F):
    """The slab projection as the path runs it (restricted to the texels the posed body can sample,
    ops.triplane_project(region=...)) and as a whole-slab stream, each timed stand-alone over 10 launches.  Algorithmic
    bytes of the region form = channels x 4 B x the texels of the frames' rectangles (whole quads, as the kernel walks
    them; the rectangle arithmetic of csrc/triplane.hip region_of restated in torch fp32)."""
    from audio_motion_avatar_amd import ops

    w_plane, _ = renderer._head_weights()
    R, C, radius = int(cfg.triplane_resolution), int(cfg.triplane_feature_dim), float(cfg.radius)
    boxes = ops.points_bbox(renderer._posed_vertices(smpl))

    def timed(region):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for _ in range(3):
            ops.triplane_project(tokens0, w_plane, R, region=region)
        ev[0].record()
        for _ in range(10):
            ops.triplane_project(tokens0, w_plane, R, region=region)
        ev[1].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / 10

    ms_region, ms_full = timed((boxes, radius)), timed(None)
    b = boxes.cpu()
    tap = lambda p: torch.floor((((p / radius).clamp(-1.0, 1.0) + 1.0) * R - 1.0) * 0.5)
    texels = 0
    for ax, ay in ((0, 1), (0, 2), (1, 2)):
        x0, x1 = tap(b[:, ax]).clamp(0, R - 1), (tap(b[:, 3 + ax]) + 1).clamp(0, R - 1)
        y0, y1 = tap(b[:, ay]).clamp(0, R - 1), (tap(b[:, 3 + ay]) + 1).clamp(0, R - 1)
        texels += float((((x1 // 4) - (x0 // 4) + 1) * 4 * (y1 - y0 + 1)).sum())
    region_bytes, slab = C * 4 * texels, F * 3 * C * R * R * 4
    return {"bound": "hbm", "kernel": "project_kernel (triplane slab, the region the posed body samples)",
            "achieved": region_bytes / (ms_region * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": region_bytes / (ms_region * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": ms_region,
            "algorithmic_bytes_per_launch": region_bytes, "traffic": None, "region_fraction_of_slab": region_bytes / slab,
            "whole_slab_stream": {"avg_launch_ms": ms_full, "achieved": slab / (ms_full * 1e-3) / 1e9,
                                  "frac": slab / (ms_full * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": slab},
            "note": "timed stand-alone in this process (10 launches each); a plane row of 32 texels is one 128-byte line "
                    "per channel, so at R = 32 only the rectangle's rows save traffic"}


def run_full_workload(args, device, world, rank, dist):
    """--workload full: configs[2] as the primary line (configs[3] with --gpus N: every rank rolls its own clip of
    `frames` frames from seeded tokens, segment parallel, SURVEY.md section 8e option i, and the uint8 frames are
    all-gathered)."""
    from audio_motion_avatar_amd.dist import FrameAllGather

    fp = FullPath(args, device, rank, args.frames)
    F, N, H, W = args.frames, args.gaussians, args.image, args.image
    gather = FrameAllGather(F, H, W, world, device, wire=args.wire, algorithm=args.exchange) if dist is not None else None
    workspaces = fp.workspaces

    def step():
        rgba = fp.step()
        if gather is not None and (gather.wire == "dense" or gather.capacity is not None):
            hint = workspaces[0].tile_counts() if gather.wire == "sparse" and len(workspaces) == 1 else None
            gather.submit(rgba, tile_hint=hint)
        return rgba

    fp.size_workspaces()
    if gather is not None:  # size the exchange buffers from one good step (host sync + MAX all-reduce, untimed)
        first = step()
        gather.calibrate(first, tile_hint=workspaces[0].tile_counts() if len(workspaces) == 1 else None)
        del first
    for _ in range(args.warmup):
        step()
    if gather is not None:
        gather.wait()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rgba = step()
    if gather is not None:
        gather.wait()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    status = [ws.status() for ws in workspaces]
    total = sum(s[0] for s in status)
    assert not any(s[1] for s in status), "rasterizer workspace overflowed inside the timed region"
    assert gather is None or not gather.overflowed(), "exchange wire buffer overflowed inside the timed region"
    result = {
        "metric": "rendered frames/sec @512x512, 10k Gaussians (audio -> Wav2Vec2 -> AudioTriplaneNet -> SMPL-X LBS -> "
                  "decode -> rasterize)",
        "value": world * F * args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": TRANSFORMER_DTYPE, "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: synthetic 16 kHz audio -> Wav2Vec2 (random weights) -> "
                               f"AudioTriplaneNet (8 layers, S=6304, {fp.windows} chained windows of {fp.T} steps) -> "
                               "SMPLXDecoder -> LBS -> decode -> rasterize 250 x 512x512",
                   "frames_per_gpu_per_step": F, "gaussians": N, "image": [H, W],
                   "weights": "random init (transformer.proj_out scaled by 0.02 so the AR chain stays bounded)",
                   "instances_per_step": int(total), "output_finite": bool(torch.isfinite(rgba).all()),
                   "exchange": exchange_description(gather)},
        "roofline": fp.attention_roofline(),
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"], result["parity"] = fp.cpu_baseline_and_parity()
    if rank == 0:
        emit(result)


def exchange_description(gather):
    if gather is None:
        return "none"
    if gather.wire == "dense":
        return {"collective": "RCCL all-gather of uint8 RGB frames", "algorithm": gather.algorithm,
                "bytes_per_rank_per_step": int(gather.local[0].numel())}
    return {"collective": "RCCL all-gather of the non-background 16x16 tiles of the uint8 RGB frames (lossless), "
                          "written in wire format by the blend kernel itself, unpacked to dense frames on every rank",
            "algorithm": gather.algorithm, "bytes_per_rank_per_step": gather.wire_bytes_per_rank(),
            "capacity_tiles": gather.capacity,
            "dense_bytes_per_rank_per_step": gather.frames * gather.height * gather.width * 3}


def host_cores():
    """Cores this process may use (the GPU box gives a 1-GPU job 16 of its 256)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline_and_parity(renderer, cfg, tokens, smpl, cam, step_outputs, n_frames):
    """Time the CPU oracle (a port of the path) on the first n_frames of the shard, and check every HIP stage
    against it on identical inputs (LBS vertices <= 1e-5, decoded Gaussians, RGB/alpha <= 1e-3)."""
    import numpy as np

    from oracle import lbs as o_lbs, rasterizer as o_rast, subdivide as o_sub, triplane as o_tri

    cores = host_cores()
    torch.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    model = renderer.smplx_model.oracle_arrays(torch.float32)
    levels = o_sub.subdivision_levels(renderer.smplx_model.faces, renderer.smplx_model.num_verts,
                                      max(1, cfg.subdivide_steps))
    idx = renderer.subset_index
    params = {"gaussian_decoder." + k: v.detach().cpu() for k, v in renderer.gaussian_decoder.state_dict().items()}
    tok = tokens[:, :n_frames].cpu()
    sp = {k: v[:, :n_frames].cpu() for k, v in smpl.items()}
    K, E = cam["intrinsic"][:, :n_frames].cpu(), cam["extrinsic"][:, :n_frames].cpu()
    planes = o_tri.tokens_to_planes(tok, cfg.triplane_resolution)

    def run():
        pts = o_lbs.get_smpl_vertices(model, sp, densify=(levels, idx))
        g = o_tri.decode_gaussians(params, planes, pts, sp["transl"].reshape(-1, 3), cfg.radius)
        return pts, g, o_rast.render_batch(g, K, E, cfg.image_size, full=True)

    run()  # warm-up (page-in, thread pools)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        pts, g, (img, alpha, unstable) = run()
        times.append(time.perf_counter() - t0)
    sec = sorted(times)[1]

    # stage-by-stage parity on identical inputs
    gpu_packed, gpu_rgba = (t[:n_frames].cpu() for t in step_outputs)
    gpu_g = renderer.unpack_gaussians(gpu_packed)
    with torch.no_grad():  # the unfused product entry point of the same stage (Renderer.get_smpl_vertices), on the
        # WHOLE shard: the frame count selects the LBS kernel (> 16 frames: the split-product MFMA kernel), and the decode
        # check below needs the points the timed step itself used
        gpu_pts = renderer.get_smpl_vertices(smpl)[:n_frames].cpu()
    g_on_gpu_pts = o_tri.decode_gaussians(params, planes, gpu_pts, sp["transl"].reshape(-1, 3), cfg.radius)
    ref_img, ref_alpha, ref_unstable = o_rast.render_batch({k: v.contiguous() for k, v in gpu_g.items()}, K, E,
                                                           cfg.image_size, full=True)
    d_rgb = (gpu_rgba[..., :3] - ref_img[0]).abs()
    d_a = (gpu_rgba[..., 3] - ref_alpha[0]).abs()
    mse = float(((gpu_rgba[..., :3] - img[0]) ** 2).mean())
    rgb_report = pixel_report(d_rgb.amax(-1), ref_unstable[0])
    alpha_report = pixel_report(d_a, ref_unstable[0])
    parity = {
        "frames": n_frames,
        "lbs_points_max_abs": float((gpu_pts - pts).abs().max()),
        "lbs_tolerance": 1e-5,
        "decode_max_abs": max(float((gpu_g[k] - g_on_gpu_pts[k]).abs().max()) for k in ("xyz", "scale", "rot",
                                                                                       "opacity", "color")),
        "decode_tolerance": 2e-5,
        "raster_rgb": rgb_report, "raster_alpha": alpha_report, "raster_tolerance": 1e-3,
        "end_to_end_psnr_db": float(10 * np.log10(1.0 / max(mse, 1e-20))),
    }
    parity["pass"] = bool(parity["lbs_points_max_abs"] <= 1e-5 and parity["decode_max_abs"] <= 2e-5 and
                          all(r["unflagged_pixels_above_tolerance"] == 0 and
                              r["pixels_above_tolerance"] <= r["pixels_flagged"] and
                              r["max_abs_all_pixels"] <= 1.2e-2 for r in (rgb_report, alpha_report)))
    base = {"value": n_frames / sec, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n_frames} frames of the same workload through oracle/ (torch CPU LBS + grid_sample/linear "
                      f"decode with {cores} threads, C rasterizer with OpenMP), median of 3 after 1 warm-up"}
    return base, parity


def emit(result):
    """The one JSON line of the contract, on the process's original stdout."""
    os.write(JSON_FD, (json.dumps(result) + "\n").encode())


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run --nproc-per-node {args.gpus}")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if os.environ.get("AMAV_BENCH_SINGLE_DEVICE") == "1":  # rehearsal of N > 1 on a one-GPU box (with the gloo backend)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    dist = None
    force_exchange = os.environ.get("AMAV_BENCH_FORCE_EXCHANGE") == "1"  # rehearse the N > 1 path on one GPU
    if world > 1 or force_exchange:
        import torch.distributed as dist

        if force_exchange and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RANK="0", WORLD_SIZE="1")
        backend = os.environ.get("AMAV_BENCH_BACKEND", "nccl")  # "gloo" only to rehearse the rank logic without xGMI
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend)

    if args.steps is None:
        args.steps = {"render": 200, "stress": 100, "full": 3}[args.workload]
    if args.warmup is None:
        args.warmup = {"render": 10, "stress": 5, "full": 1}[args.workload]
    if args.workload == "stress":
        args.gaussians, args.image = 50000, 1024
        if args.frames == 250:
            args.frames = 32  # 3.2 GB of triplane tokens per step
        args.cpu_frames = min(args.cpu_frames, 1)
    if args.workload == "full":
        run_full_workload(args, device, world, rank, dist)
        if dist is not None:
            dist.destroy_process_group()
        return

    from audio_motion_avatar_amd import ops
    from audio_motion_avatar_amd.dist import FrameAllGather
    from audio_motion_avatar_amd.synthetic import make_render_inputs

    renderer, cfg = build_renderer(args, device)
    F, N, H, W = args.frames, args.gaussians, args.image, args.image
    tokens, smpl, cam = make_render_inputs(F, cfg, seed=42 + rank, device=device)
    smpl_tokens = torch.zeros(1, F, 1, 1, device=device)  # only its [B,T] shape is read when no decoder is attached
    workspaces = [None] * max(1, min(args.chunks, F))  # filled by the first step, then reused
    gather = FrameAllGather(F, H, W, world, device, wire=args.wire, algorithm=args.exchange) if dist is not None else None

    stages = [None, None]  # the last step's intermediate tensors (parity check)

    def step():
        # the body of Renderer.forward (renderer.py:73-204) with the rasterizer workspace pinned and its overflow
        # check (the only host sync) deferred to the end of the run; returns the RGBA buffer [1,F,H,W,4]
        rgba, packed = renderer.render_tokens(tokens[0], smpl, cam, chunks=args.chunks, workspaces=workspaces,
                                              check_overflow=False)
        stages[:] = [packed, rgba]
        return rgba.unsqueeze(0)

    # one eager step: validates the drop-in entry point end to end and sizes the workspace
    with torch.no_grad():
        ref_img = renderer(tokens, cam, smpl_tokens, smpl)[0]
        rgba = step()
        torch.cuda.synchronize()
        for ci, ws in enumerate(workspaces):
            _, max_frame, over = ws.status_full()
            if over:
                fc, n, h, w = ws.key
                workspaces[ci] = ops.RasterWorkspace(fc, n, h, w, int(fc * max_frame * 1.25), device)
        rgba = step()
        assert not any(ws.status()[1] for ws in workspaces)
        assert torch.equal(ref_img, rgba[..., :3]), "pinned-workspace step differs from Renderer.forward"
        if gather is not None:
            # sizes the wire buffers (host sync + MAX all-reduce, outside the timed region)
            gather.calibrate(rgba, tile_hint=workspaces[0].tile_counts() if len(workspaces) == 1 else None)
        del ref_img, rgba

    nchunks = len(workspaces)
    events = [[(ops.Event(), ops.Event()) for _ in range(nchunks)] for _ in range(args.steps)]

    # N > 1, sparse wire, one frame group: the rasterizer writes the exchange's wire buffer itself (no pack pass)
    direct_wire = gather is not None and gather.wire == "sparse" and len(workspaces) == 1 and os.environ.get("AMAV_WIRE_PACK") != "1"

    def timed_step(i):
        ops.PROFILE_EVENTS = list(events[i]) if i is not None else None
        with torch.no_grad():
            if direct_wire:
                rgba, packed = renderer.render_tokens(tokens[0], smpl, cam, chunks=1, workspaces=workspaces,
                                                      check_overflow=False, wire=gather.wire_target())
                stages[:] = [packed, rgba]
                out = rgba.unsqueeze(0)
            else:
                out = step()
        if gather is not None:  # (pack +) RCCL all-gather + unpack on a side stream, overlapping the next step
            hint = workspaces[0].tile_counts() if gather.wire == "sparse" and len(workspaces) == 1 and not direct_wire else None
            gather.submit(out, tile_hint=hint, packed=direct_wire)
        return out

    for _ in range(args.warmup):
        timed_step(None)
    if gather is not None:
        gather.wait()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()  # (twice: whatever the collective library sets up lazily for a barrier is paid here, not at
        torch.cuda.synchronize()  # the barrier that closes the timed region)
        dist.barrier()
    step_marks = [ops.Event() for _ in range(args.steps + 1)]  # device-side duration of every step
    t0 = time.perf_counter()
    step_marks[0].record()
    # Opt-in experiment (not the reported configuration): consecutive steps on alternating streams, each with its own
    # rasterizer workspace, so that the next step's LBS / projection start in the tail of the blend kernel, where fewer
    # than half of its waves still hold a tile.  Measured: 0.838 -> 0.812 ms per step (the blend kernel itself 0.42 ->
    # 0.44 ms).  The default keeps one stream: the step-by-step device times and the N > 1 exchange are defined on it.
    _nstreams = int(os.environ.get("AMAV_BENCH_STEP_STREAMS", "1"))
    if _nstreams > 1:
        _streams = [torch.cuda.Stream(device=device) for _ in range(_nstreams)]
        _ws_sets = [list(workspaces)] + [[ops.RasterWorkspace(*w.key, w.capacity, device) for w in workspaces] for _ in range(_nstreams - 1)]
        for st in _streams:
            st.wait_stream(torch.cuda.current_stream())
    for i in range(args.steps):
        if _nstreams > 1:
            workspaces[:] = _ws_sets[i % _nstreams]
            with torch.cuda.stream(_streams[i % _nstreams]):
                out = timed_step(i)
        else:
            out = timed_step(i)
        step_marks[i + 1].record()
    _dbg = os.environ.get("AMAV_BENCH_DEBUG") == "1"
    _t1 = time.perf_counter()
    if gather is not None:
        gather.wait()
    torch.cuda.synchronize()
    _t2 = time.perf_counter()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if _dbg:
        print(f"[debug] enqueue {(_t1 - t0) * 1e3:.2f} ms, drain {(_t2 - _t1) * 1e3:.2f} ms, barrier {(time.perf_counter() - _t2) * 1e3:.2f} ms", file=sys.stderr)
    ops.PROFILE_EVENTS = None
    step_ms = sorted(step_marks[i].elapsed_ms(step_marks[i + 1]) for i in range(args.steps))
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    status = [ws.status() for ws in workspaces]
    total = sum(s[0] for s in status)
    assert not any(s[1] for s in status), "rasterizer workspace overflowed inside the timed region"
    assert gather is None or not gather.overflowed(), "exchange wire buffer overflowed inside the timed region"
    exchange_ok = None
    if gather is not None:  # the reassembled clip of the last step: this rank's block must be its own frames
        full = gather.full[gather.turn ^ 1]
        mine = full[rank * F:(rank + 1) * F]
        ok = torch.equal(mine, ops.frames_to_rgb8(out.view(F, H, W, 4)))
        others = [full[r * F:(r + 1) * F] for r in range(world) if r != rank]
        ok = ok and all(bool((o != 255).any()) for o in others)  # and the other blocks hold rendered frames
        t = torch.tensor([int(ok)], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        exchange_ok = bool(t.item())
        assert exchange_ok, "the all-gathered clip does not match the frames the ranks rendered"

    # blend-kernel time per step = sum over the step's frame groups (one launch each)
    blend_ms = sorted(sum(s.elapsed_ms(e) for s, e in step_events) for step_events in events)
    blend_avg_ms = sum(blend_ms) / len(blend_ms)
    # algorithmic bytes of the blend kernel per frame: RGBA out (16 B/px) + one 40 B record per Gaussian
    # (xy, conic, opacity, rgb, depth), each moved once (DESIGN.md "kernels")
    blend_bytes = F * (16 * H * W + 40 * N)
    achieved = blend_bytes / (blend_avg_ms * 1e-3) / 1e9
    result = {
        "metric": "rendered frames/sec @512x512, 10k Gaussians (static triplane decode + SMPL-X LBS + tile rasterize)",
        "value": world * F * args.steps / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 (LBS blend-shape product: fp16 x 2 split operands with fp32 accumulation, fp32-equivalent; "
                 "everything else plain fp32)",
        "data": "synthetic",
        "step_device_ms": {"min": step_ms[0], "median": step_ms[len(step_ms) // 2], "max": step_ms[-1],
                           "note": "HIP events between consecutive steps on the compute stream (the timed region is short)"},
        "config": {"workload": ("BASELINE configs[4] (per GPU): triplane 128^2 x 512 ch, 50k Gaussians, 1024x1024, static "
                                "triplane decode + LBS + rasterize, no audio net" if args.workload == "stress" else
                                "BASELINE configs[1]: 512x512, 10k Gaussians, static triplane decode + LBS + rasterize, "
                                "no audio net"), "frames_per_gpu_per_step": F, "gaussians": N, "image": [H, W],
                   "triplane": [cfg.triplane_feature_dim, cfg.triplane_resolution],
                   "instances_per_step": int(total), "stream_pipelined_frame_groups": len(workspaces),
                   "exchange": exchange_description(gather), "exchange_verified": exchange_ok},
        "roofline": {"bound": "hbm", "kernel": "render_kernel (tile blend)", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": pmc_traffic_bytes("render_kernel") if (F, N, H) == (250, 10000, 512) else None,
                     "avg_launch_ms": blend_avg_ms, "algorithmic_bytes_per_launch": blend_bytes,
                     "issue": pmc_issue_fractions("render_kernel") if (F, N, H) == (250, 10000, 512) else None},
    }
    if args.workload == "stress":  # the stage that dominates this configuration is the slab projection (HBM stream)
        with torch.no_grad():
            result["roofline_project"] = project_roofline(renderer, cfg, tokens[0], smpl, F)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base, parity = cpu_baseline_and_parity(renderer, cfg, tokens, smpl, cam, stages, args.cpu_frames)
        result["cpu_baseline"] = base
        result["parity"] = parity
    if rank == 0 and world == 1 and args.workload == "render" and not args.no_full_path:
        torch.cuda.empty_cache()
        result["full_path"], fp = measure_full_path(args, device, rank, args.full_steps, 1, not args.no_cpu_baseline)
        result["full_path_exact_fp32"] = measure_full_path_exact_fp32(fp, result["full_path"], 1, not args.no_cpu_baseline)
        del fp
    if rank == 0 and world == 1 and args.workload == "render" and not args.no_refiner and N == 10000:
        torch.cuda.empty_cache()
        result["point_refiner"] = measure_point_refiner(args, device, not args.no_cpu_baseline)
    if rank == 0 and world == 1 and args.workload == "render" and not args.no_extra_configs:
        torch.cuda.empty_cache()
        result["reference_defaults"] = measure_reference_defaults(args, device, not args.no_cpu_baseline)
        fpath = result.get("full_path")
        if fpath:  # the reference's default configuration behind the audio net: one transformer step per frame
            step_ms = fpath["roofline"]["transformer_step"]["ms"]
            result["reference_defaults"]["with_transformer_step_frames_per_s"] = 1e3 / (result["reference_defaults"]["ms_per_frame"] + step_ms)
        torch.cuda.empty_cache()
        result["stress"] = measure_stress(args, device, not args.no_cpu_baseline)
    if rank == 0:
        emit(result)
    if dist is not None:
        dist.destroy_process_group()
    # a parity object that exists must carry an explicit `pass`: a missing flag is a failure, not a success
    failed = [k for k in ("parity",) if k in result and result[k].get("pass") is not True]
    failed += [k for k in EXTRA_OBJECTS if "parity" in result.get(k, {}) and result[k]["parity"].get("pass") is not True]
    if rank == 0 and failed:
        raise SystemExit(f"parity check against the CPU oracle FAILED in {failed} (see the JSON line)")


if __name__ == "__main__":
    # RCCL prints its version banner on stdout when the communicator is created: keep the original stdout for the JSON
    # line only and send everything else (C libraries included) to stderr
    JSON_FD = os.dup(1)
    os.dup2(2, 1)
    main()
