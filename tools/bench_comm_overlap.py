#!/usr/bin/env python3
"""Diagnostic: can a collective-like kernel (tools/comm_emulator.hip: 32 workgroups x 256 threads, ~100 VGPRs, resident
for 1 ms) run NEXT TO the render step, or does the persistent blend kernel (which owns every wave slot at 20 waves per
CU) make it wait?  Step time of the 250-frame render workload with the emulator on a side stream, for several
AMAV_RENDER_WAVES (resident blend waves per CU).  Build the emulator on the box first (see its header).

    python tools/bench_comm_overlap.py <waves per CU> [emulator microseconds]
"""
import ctypes
import os
import sys
import time

waves = sys.argv[1] if len(sys.argv) > 1 else "20"
micro = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
os.environ["AMAV_RENDER_WAVES"] = waves
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_motion_avatar_amd.config import RendererConfig  # noqa: E402
from audio_motion_avatar_amd.renderer import Renderer  # noqa: E402
from audio_motion_avatar_amd.synthetic import init_random_heads, make_render_inputs  # noqa: E402

emu = ctypes.CDLL(os.path.join(ROOT, "gpurun_out", "libcomm_emulator.so"))
emu.comm_emulate.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
F, H, W = 250, 512, 512
cfg = RendererConfig(image_size=(H, W), subdivide_steps=0, predict_smplx_params=False, device="cuda")
r = init_random_heads(Renderer(cfg).eval())
tokens, smpl, cam = make_render_inputs(F, cfg, seed=42)
ws = [None]
side = torch.cuda.Stream()
sink = torch.zeros(4, device="cuda")


def run(steps, with_comm):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        rgba, _ = r.render_tokens(tokens[0], smpl, cam, workspaces=ws, check_overflow=False)
        if with_comm:  # like FrameAllGather.submit: the side stream waits for this step's frames, then "communicates"
            side.wait_stream(torch.cuda.current_stream())
            emu.comm_emulate(32, micro, sink.data_ptr(), side.cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


with torch.no_grad():
    run(5, False)
    a = run(40, False)
    run(5, True)
    b = run(40, True)
print(f"AMAV_RENDER_WAVES={waves}: step {a:.3f} ms alone, {b:.3f} ms with a {micro} us resident side-stream kernel per step")
