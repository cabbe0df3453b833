"""The N > 1 exchange end to end on the GPU box: two processes (gloo over localhost, both on cuda:0 -- RCCL needs one
GPU per rank) render different shards, pack them with the HIP kernels, all-gather the wire buffers and unpack; every
rank must end up with both shards' frames, for the sparse and the dense wire format.  What this cannot cover is RCCL /
xGMI itself (the driver's multi-GPU run does)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from audio_motion_avatar_amd import ops
        from audio_motion_avatar_amd.config import RendererConfig
        from audio_motion_avatar_amd.dist import FrameAllGather
        from audio_motion_avatar_amd.renderer import Renderer
        from audio_motion_avatar_amd.synthetic import init_random_heads, make_render_inputs

        F, H, W = 6, 128, 160
        cfg = RendererConfig(image_size=(H, W), subdivide_steps=0, predict_smplx_params=False, device="cuda")
        r = init_random_heads(Renderer(cfg).eval())
        gens = []  # two generations of every rank's shard: the body moves, so tiles appear and disappear between them
        with torch.no_grad():
            for gen in range(2):
                shards = []
                for rk in range(world):  # every rank renders every shard, so it knows what the gather must deliver
                    tokens, smpl, cam = make_render_inputs(F, cfg, seed=100 + rk + 50 * gen, device="cuda")
                    smpl["transl"][..., 0] += 0.35 * gen
                    ws = [None]
                    rgba, _ = r.render_tokens(tokens[0], smpl, cam, workspaces=ws)
                    shards.append((rgba.clone(), ws[0].tile_counts().clone(), (tokens[0], smpl, cam)))
                gens.append((shards, torch.cat([ops.frames_to_rgb8(s[0]) for s in shards])))
        assert not torch.equal(gens[0][1], gens[1][1])
        ok = {}
        for wire, algorithm in (("sparse", "collective"), ("dense", "collective"), ("sparse", "direct"), ("dense", "direct")):
            gather = FrameAllGather(F, H, W, world, "cuda", wire=wire, algorithm=algorithm)
            if wire == "sparse":
                gather.calibrate(gens[0][0][rank][0], headroom=2.0, tile_hint=gens[0][0][rank][1])
                assert gather.tile_state[0] is not None  # W % 16 == 0: the differential unpack is what runs
            # both buffers of the double buffering, reuses with the same frames, and reuses after the frames changed
            # (the differential unpack must re-clear the tiles the body left)
            for gen in (0, 0, 0, 1, 1, 0, 1, 0, 0):
                shards, want = gens[gen]
                mine, hint, _ = shards[rank]
                full = gather.submit(mine, tile_hint=hint if wire == "sparse" else None)
                gather.wait()
                torch.cuda.synchronize()
                key = f"{wire}/{algorithm}"
                ok[key] = bool(torch.equal(full, want)) and not gather.overflowed() and ok.get(key, True)
            if wire == "sparse":
                # the same steps with the wire buffer written by the blend kernel itself (no pack pass)
                for gen in (0, 1, 1, 0, 0, 1):
                    shards, want = gens[gen]
                    tok, smpl_g, cam_g = shards[rank][2]
                    with torch.no_grad():
                        rgba, _ = r.render_tokens(tok, smpl_g, cam_g, wire=gather.wire_target())
                    full = gather.submit(rgba, packed=True)
                    gather.wait()
                    torch.cuda.synchronize()
                    key = f"{wire}/{algorithm}/rasterizer-wire"
                    ok[key] = bool(torch.equal(full, want)) and not gather.overflowed() and ok.get(key, True)
        flags = [None] * world
        dist.all_gather_object(flags, ok)
        if rank == 0:
            out.put(flags)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_exchange_their_shards():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        flags = q.get(timeout=400)
    finally:
        for p in procs:
            p.join(timeout=400)
    assert all(p.exitcode == 0 for p in procs)
    assert flags == [{"sparse/collective": True, "dense/collective": True, "sparse/direct": True, "dense/direct": True,
                      "sparse/collective/rasterizer-wire": True, "sparse/direct/rasterizer-wire": True}] * 2


def _harness_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from audio_motion_avatar_amd import ops
        from audio_motion_avatar_amd.harness import AudioDrivenAvatar
        from audio_motion_avatar_amd.synthetic import init_random_heads, make_render_inputs
        from test_audio_net_gpu import randomize, small_cfg

        cfg = small_cfg()
        torch.manual_seed(1234)  # every parameter the constructors draw: identical replicas on both ranks
        model = AudioDrivenAvatar(cfg)
        randomize(model.audio_triplane.transformer, 21)
        init_random_heads(model.renderer)
        model = model.cuda()
        T, W = 3, 2
        _, _, cam = make_render_inputs(T * W, cfg.renderer, seed=8, batch=1)
        g = torch.Generator().manual_seed(3)
        audio = torch.randn(1, T * W, 48, generator=g).cuda()
        tri = torch.randn(1, 2, 32, 192, generator=g).cuda()
        smpl = (torch.randn(1, 2, 32, 10, generator=g) * 0.2).cuda()
        ref = model.rollout(tri, smpl, audio, cam)["images"][0]  # the exact chain, on this rank alone
        want = ops.frames_to_rgb8(torch.cat([ref, torch.ones_like(ref[..., :1])], dim=-1).contiguous())
        seq = model.rollout_sharded(tri, smpl, audio, cam, mode="sequential")
        seg = model.rollout_sharded(tri, smpl, audio, cam, mode="segment")
        # segment mode restarts rank 1's window from the reference tokens: its first window equals the chain's first
        ok = {"sequential": bool(torch.equal(seq, want)),
              "segment_rank0_block": bool(torch.equal(seg[:T], want[:T])),
              "segment_rank1_block_is_a_fresh_chain": bool(seg.shape == want.shape and not torch.equal(seg[T:], want[T:]))}
        flags = [None] * world
        dist.all_gather_object(flags, ok)
        if rank == 0:
            out.put(flags)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_sequential_and_segment_rollout():
    """harness.rollout_sharded on two ranks: "sequential" (rank 0 runs the one chain and sends rank 1 its token block)
    reproduces the single-GPU clip on both ranks; "segment" gives rank 1 a chain of its own."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_harness_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        flags = q.get(timeout=400)
    finally:
        for p in procs:
            p.join(timeout=400)
    assert all(p.exitcode == 0 for p in procs)
    assert flags == [{"sequential": True, "segment_rank0_block": True, "segment_rank1_block_is_a_fresh_chain": True}] * 2
