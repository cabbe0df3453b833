#!/usr/bin/env python3
"""Diagnostic: where do the GPU path and the CPU oracle part ways on the reference's default renderer configuration?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_motion_avatar_amd import ops
from audio_motion_avatar_amd.config import RendererConfig
from audio_motion_avatar_amd.renderer import Renderer
from audio_motion_avatar_amd.synthetic import init_random_heads, make_render_inputs
from oracle import lbs as o_lbs, subdivide as o_sub, triplane as o_tri

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = RendererConfig(image_size=(512, 512), subdivide_steps=steps, upsample_triplane=True, no_point_refiner=False,
                     predict_smplx_params=False, device="cuda")
torch.manual_seed(11)
r = init_random_heads(Renderer(cfg).eval())
with torch.no_grad():
    r.point_refiner[-1].weight.normal_(0, 0.005)
    tokens, smpl, cam = make_render_inputs(1, cfg, seed=42)
    up_tokens = r.triplane_upsampler.forward_tokens(tokens[0], cfg.triplane_resolution)       # full planes on the GPU
    verts = r.get_smpl_vertices(smpl)
    refined = r.refine_points(up_tokens, verts)
    params = {k: v.detach().cpu() for k, v in r.state_dict().items()}
    sp = {k: v.cpu() for k, v in smpl.items()}
    levels = o_sub.subdivision_levels(r.smplx_model.faces, r.smplx_model.num_verts, max(1, cfg.subdivide_steps))
    pts = o_lbs.get_smpl_vertices(r.smplx_model.oracle_arrays(torch.float32), sp, densify=(levels, r.subset_index))
    print("points", float((verts.cpu() - pts).abs().max()))
    planes = o_tri.tokens_to_planes(tokens.cpu(), cfg.triplane_resolution)
    up = o_tri.triplane_upsampler(params, planes, cfg.num_upsample_blocks)
    R = up.shape[-1]
    up_gpu = up_tokens.cpu().view(1, 256, 3, R, R).permute(0, 2, 1, 3, 4)
    print("upsampler max abs diff", float((up_gpu - up).abs().max()), "scale", float(up.abs().max()))
    pcfg = {k: list(getattr(cfg, k)) for k in ("enc_depths", "enc_num_head", "enc_patch_size", "dec_depths", "dec_num_head", "dec_patch_size")}
    f_cpu = o_tri.sample_from_triplane(up, pts, cfg.radius)
    f_gpu = ops.triplane_sample_features(up_tokens.view(1, 256, 3, R, R).permute(0, 2, 1, 3, 4), verts, cfg.radius).cpu()
    print("features max abs diff", float((f_cpu - f_gpu).abs().max()), "scale", float(f_cpu.abs().max()))
    want_same_planes = o_tri.refine_points(params, up_gpu.contiguous(), pts, cfg.radius, pcfg)   # oracle on the GPU's planes
    print("refined (oracle on GPU planes) vs GPU", float((refined.cpu() - want_same_planes).abs().max()),
          "largest offset", float((want_same_planes - pts).abs().max()))
    same_pts = o_tri.refine_points(params, up_gpu.contiguous(), verts.cpu(), cfg.radius, pcfg)  # ... and on the GPU's points
    print("refined (oracle on GPU planes AND GPU points) vs GPU", float((refined.cpu() - same_pts).abs().max()))
    g100 = torch.floor(100 * verts.cpu()); c100 = torch.floor(100 * pts)
    print("points whose voxel differs between the two LBS results:", int((g100 != c100).any(-1).sum()))
    want = o_tri.refine_points(params, up, pts, cfg.radius, pcfg)
    print("refined (oracle on CPU planes) vs GPU", float((refined.cpu() - want).abs().max()))
    print("oracle: CPU planes vs GPU planes", float((want - want_same_planes).abs().max()))
