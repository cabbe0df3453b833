#!/usr/bin/env python3
"""Benchmarks the library's GEMM candidates (PyTorch TunableOp: hipBLASLt + rocBLAS solutions) for the fixed GEMM
shapes of the path on THIS GPU and writes audio-motion-avatar_amd/gemm_tuning_gfx950.csv (looked up at run time by
audio_motion_avatar_amd.tuning; tuning itself never runs in the product).  Workloads: one 250-frame clip of the full
audio-driven path (transformer step at S = 6304, reducers, SMPL-X decoder, Wav2Vec2 on a 10 s clip) and one pass of
the point refiner at 8 x 10 000 points (only its fixed-size level-0 shapes are kept).

    AMAV_TUNED_GEMMS=tune python tools/tune_gemms.py [out.csv]
"""
import os
import sys

os.environ["AMAV_TUNED_GEMMS"] = "tune"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.cuda.tunable as tunable  # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "audio-motion-avatar_amd", "gemm_tuning_gfx950.csv")
tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_filename(os.path.join(ROOT, "gpurun_out", "tunableop_scratch.csv"))
tunable.set_max_tuning_duration(30)
tunable.set_max_tuning_iterations(50)

sys.argv = ["bench.py", "--workload", "full", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
import bench  # noqa: E402

args = bench.parse()
with torch.no_grad():
    fp = bench.FullPath(args, "cuda:0", 0, args.frames)
    fp.size_workspaces()
    fp.step()
    torch.cuda.synchronize()
fixed = {(r[0], r[1]) for r in tunable.get_results()}
print(f"full path: {len(fixed)} GEMM shapes tuned", flush=True)
del fp
torch.cuda.empty_cache()

from audio_motion_avatar_amd import ops  # noqa: E402
from audio_motion_avatar_amd.config import RendererConfig  # noqa: E402
from audio_motion_avatar_amd.renderer import Renderer  # noqa: E402
from audio_motion_avatar_amd.synthetic import init_random_heads, make_render_inputs  # noqa: E402

per = 8
cfg = RendererConfig(image_size=(512, 512), subdivide_steps=0, predict_smplx_params=False, no_point_refiner=False,
                     refiner_clouds_per_pass=per, device="cuda")
r = init_random_heads(Renderer(cfg).eval())
tokens, smpl, _ = make_render_inputs(per, cfg, seed=42)
with torch.no_grad():
    verts = ops.points_gather(r._posed_vertices(smpl), r._gather_idx)
    r.refine_points(tokens[0], verts)
    torch.cuda.synchronize()
rows = []
level0 = f"_{per * verts.shape[1]}_"
for op, key, solution, ms in tunable.get_results():
    if (op, key) in fixed or level0 in key:
        rows.append((op, key, solution, ms))
print(f"point refiner: kept {len(rows) - len(fixed)} level-0 shapes of {len(tunable.get_results()) - len(fixed)}", flush=True)
with open(out_path, "w") as fh:
    for v in tunable.get_validators():
        fh.write("Validator," + ",".join(str(x) for x in v) + "\n")
    for op, key, solution, ms in sorted(rows):
        fh.write(f"{op},{key},{solution},{ms:.6g}\n")
print("wrote", out_path, len(rows), "entries")
