"""GPU parity of the HIP tile rasterizer (through the C ABI) against the CPU oracle.

Tolerance (BASELINE.json north_star): RGB / alpha max-abs <= 1e-3 against the fp32 oracle.  The algorithm itself is
discontinuous (skip when alpha < 1/255, stop when T' < 1e-4, skip when power > 0), so two correct fp32
implementations can take different branches where a value sits within rounding of a threshold.  The oracle flags
those pixels (`unstable`: a decision within 1e-4 relative of a threshold).  Every comparison
prints (and logs to gpurun_out/raster_parity.jsonl) the maximum over ALL pixels, the number of pixels above 1e-3 and
the number flagged, and holds
    * every unflagged pixel to 1e-3 (measured differences are ~1e-6),
    * the number of pixels above 1e-3 to at most the number flagged,
    * every flagged pixel to one flipped decision (<= 1.2e-2), and
    * the flagged set to < 0.5 % of the image,
while geometry decisions (radii, tile rectangles, depth order, instance counts) must match exactly: the HIP
preprocess uses the oracle's operation order with contraction off.
"""
import numpy as np
import pytest
import torch

from helpers import oracle_frames, random_scene

pytestmark = pytest.mark.gpu
TOL = 1e-3
FLIP_TOL = 1.2e-2


def run_hip(scene, **kw):
    from audio_motion_avatar_amd import ops

    dev = "cuda"
    view, proj, tanfov, _ = ops.camera_from_intrinsics(scene["K"].to(dev), scene["E"].to(dev), scene["H"], scene["W"])
    c = lambda k: scene[k].to(dev)
    out = ops.rasterize(c("xyz"), c("rot"), c("scale"), c("opacity"), c("color"), view, proj, tanfov, scene["H"],
                        scene["W"], want_inv_depth=True, want_radii=True, **kw)
    torch.cuda.synchronize()
    return out


def report(test, frame, name, diff, unstable, tol=TOL):
    """All-pixel statistics of one comparison (printed, and appended to gpurun_out/raster_parity.jsonl when that
    directory exists): maximum over ALL pixels, pixels above the tolerance, pixels the oracle flags as sitting on one
    of the algorithm's discontinuities.  Fails when an UNFLAGGED pixel is off by more than the tolerance, when more
    pixels are off than were flagged, or when a flagged pixel is off by more than one flipped decision."""
    import json
    import os

    above = diff > tol
    rec = {"test": test, "frame": int(frame), "what": name, "pixels": int(diff.size), "max_abs_all_pixels": float(diff.max()),
           "max_abs_unflagged": float((diff * ~unstable).max()), "pixels_above_tol": int(above.sum()),
           "unflagged_above_tol": int((above & ~unstable).sum()), "pixels_flagged": int(unstable.sum()), "tol": tol}
    print("raster parity:", json.dumps(rec))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "raster_parity.jsonl"), "a") as fh:
            fh.write(json.dumps(rec) + "\n")
    assert rec["unflagged_above_tol"] == 0, f"{name}: unflagged pixels off by {rec['max_abs_unflagged']}"
    assert rec["pixels_above_tol"] <= rec["pixels_flagged"], f"{name}: {rec['pixels_above_tol']} pixels off, {rec['pixels_flagged']} flagged"
    assert rec["max_abs_all_pixels"] <= FLIP_TOL, f"{name}: flagged pixels off by {rec['max_abs_all_pixels']}"
    return rec


def compare(scene, out, tol=TOL, **settings):
    import inspect

    test = next((fr.function for fr in inspect.stack() if fr.function.startswith("test_")), "?")
    ref = oracle_frames(scene, np.float32, **settings)
    rgba = out["rgba"].cpu().numpy()
    for f, r in enumerate(ref):
        got_rgb = np.moveaxis(rgba[f, :, :, :3], -1, 0)
        unstable = r["unstable"] != 0
        assert unstable.mean() < 0.005, f"frame {f}: {unstable.mean():.4%} of the pixels sit on a threshold"
        report(test, f, "rgb", np.abs(got_rgb - r["color"]).max(axis=0), unstable, tol)
        report(test, f, "alpha", np.abs(rgba[f, :, :, 3] - r["alpha"]), unstable, tol)
        report(test, f, "inv_depth", np.abs(out["inv_depth"][f].cpu().numpy() - r["inv_depth"]), unstable, tol)
        assert np.array_equal(out["radii"][f].cpu().numpy(), r["radii"]), f"frame {f} radii"
    total, over = out["workspace"].status()
    assert not over
    assert total == sum(r["instances"] for r in ref)
    return ref


@pytest.mark.parametrize("N,H,W,F", [(300, 64, 80, 1), (2000, 256, 256, 2), (500, 50, 70, 3), (64, 16, 16, 1),
                                     (150, 48, 64, 40),   # 40 frames: binning in 7 slices per frame (bin_slices)
                                     (90, 32, 48, 97)])   # >= 96 frames: the fused per-frame binning block
def test_random_scenes(N, H, W, F):
    scene = random_scene(1234 + N, N, H, W, F)
    compare(scene, run_hip(scene))


@pytest.mark.parametrize("settings", [dict(antialiasing=True), dict(scale_modifier=0.6),
                                      dict(antialiasing=True, scale_modifier=1.7)])
def test_rasterization_settings(settings):
    """GaussianRasterizationSettings.antialiasing (opacity scaled by sqrt(det / det_lowpass)) and .scale_modifier: the
    reference passes False / 1.0 (renderer.py:520,529); the op supports the other values like upstream."""
    scene = random_scene(4321, 1500, 160, 208, 2)
    compare(scene, run_hip(scene, **settings), **settings)


def test_full_size_config():
    """BASELINE configs[1] size: 512x512, 10k Gaussians (one frame is enough for the oracle to stay fast)."""
    scene = random_scene(7, 10000, 512, 512, 2, spread=0.3, log_scale=-4.9, scale_jitter=0.55)
    compare(scene, run_hip(scene))


def test_big_tiles_take_the_block_sort():
    """Thousands of large Gaussians on a 32x32 image: every tile list exceeds the per-wave LDS sort."""
    scene = random_scene(99, 3000, 32, 32, 1, spread=0.05, log_scale=-2.5, scale_jitter=0.2)
    out = run_hip(scene)
    ref = compare(scene, out)
    assert ref[0]["instances"] > 4 * 1024


def test_everything_culled_gives_background():
    scene = random_scene(5, 100, 48, 48, 1)
    scene["xyz"][..., 2] = -1.0  # behind the camera
    out = run_hip(scene, bg=(0.25, 0.5, 0.75))
    rgba = out["rgba"].cpu()
    assert torch.allclose(rgba[..., :3], torch.tensor([0.25, 0.5, 0.75]).expand_as(rgba[..., :3]))
    assert torch.count_nonzero(rgba[..., 3]) == 0
    assert out["workspace"].status() == (0, False)


def test_overflow_is_detected_and_retried():
    from audio_motion_avatar_amd import ops

    scene = random_scene(11, 800, 64, 64, 2, log_scale=-3.0)
    ws = ops.RasterWorkspace(2, 800, 64, 64, instance_capacity=10, device="cuda")
    out = run_hip(scene, workspace=ws)          # retried with an exact-size workspace
    assert out["workspace"] is not ws
    compare(scene, out)
    out2 = run_hip(scene, workspace=ws, check_overflow=False)
    total, over = ws.status()
    assert over and total > 10


def test_shared_gaussians_across_views():
    """frame_stride = 0: one Gaussian set, several cameras (render_multi_view, renderer.py:431-445)."""
    scene = random_scene(3, 400, 64, 64, 3)
    for k in ("xyz", "rot", "scale", "opacity", "color"):
        scene[k] = scene[k][:1].expand(3, -1, -1)
    compare(scene, run_hip(scene))


def test_activations_and_clamp_match_render_one():
    """apply_activations=1 + clamp_output=1 == renderer.py:532-547,568 around the rasterizer."""
    from audio_motion_avatar_amd import ops
    from oracle import rasterizer as orc

    g = torch.Generator().manual_seed(21)
    N, H, W = 1500, 96, 96
    gauss = dict(xyz=torch.randn(1, N, 3, generator=g) * 0.3 + torch.tensor([0, 0, 2.4]),
                 rot=torch.nn.functional.normalize(torch.randn(1, N, 4, generator=g), dim=-1),
                 scale=torch.randn(1, N, 3, generator=g) * 0.5,           # raw: exp(s - 3.9), capped at 0.1
                 opacity=torch.randn(1, N, 1, generator=g),
                 color=torch.rand(1, N, 3, generator=g) * 1.4 - 0.2)     # exercises the clamp
    K = torch.tensor([[[96.0, 0, 48], [0, 96.0, 48], [0, 0, 1]]])
    E = torch.eye(4)[None]
    ref, ref_alpha, unstable = orc.render_batch(gauss, K[None], E[None], (H, W), full=True)
    view, proj, tanfov, _ = ops.camera_from_intrinsics(K.cuda(), E.cuda(), H, W)
    out = ops.rasterize(*[gauss[k].cuda() for k in ("xyz", "rot", "scale", "opacity", "color")], view, proj, tanfov,
                        H, W, apply_activations=True, clamp_output=True)
    rgba = out["rgba"].cpu()
    flagged = unstable[0, 0].numpy()
    report("test_activations_and_clamp_match_render_one", 0, "rgb",
           (rgba[0, :, :, :3] - ref[0, 0]).abs().amax(-1).numpy(), flagged)
    report("test_activations_and_clamp_match_render_one", 0, "alpha", (rgba[0, :, :, 3] - ref_alpha[0, 0]).abs().numpy(),
           flagged)


def test_stress_config_one_frame():
    """BASELINE configs[4] geometry: 1024x1024, 50k Gaussians (one frame keeps the oracle in seconds)."""
    scene = random_scene(77, 50000, 1024, 1024, 1, spread=0.45, log_scale=-5.2, scale_jitter=0.6)
    compare(scene, run_hip(scene))


def test_ted_image_size_non_square_partial_tiles():
    """The reference's own frame size is 1296 x 2304 (ted_speech.yaml:14): 81 x 144 tiles; here a quarter of it,
    with a width and height that are not multiples of 16."""
    scene = random_scene(78, 4000, 325, 577, 1, spread=0.5, log_scale=-4.0)
    compare(scene, run_hip(scene))


def test_full_ted_frame_runs():
    from audio_motion_avatar_amd import ops

    scene = random_scene(79, 3000, 1296, 2304, 1, spread=0.6, log_scale=-4.2, focal=2304.0)
    out = run_hip(scene)
    ref = oracle_frames(scene, np.float32)[0]
    rgba = out["rgba"].cpu().numpy()[0]
    report("test_full_ted_frame_runs", 0, "rgb", np.abs(np.moveaxis(rgba[..., :3], -1, 0) - ref["color"]).max(axis=0),
           ref["unstable"] != 0)
    assert np.array_equal(out["radii"][0].cpu().numpy(), ref["radii"])
