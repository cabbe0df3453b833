"""The N > 1 path on CPU: two gloo ranks shard a clip and reassemble it with the same all-gather the GPU path uses."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, total_frames, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from audio_motion_avatar_amd.dist import all_gather_frames, shard_range

    s, e = shard_range(total_frames, world, rank)
    # every "rendered frame" carries its global frame id, so the reassembled order is checkable
    local = torch.arange(s, e, dtype=torch.uint8).view(-1, 1, 1, 1).expand(-1, 4, 6, 3).contiguous()
    full = all_gather_frames(local)
    # token hand-out of the "sequential" mode: rank 0 holds the whole clip's tokens, blocks differ by one frame
    from audio_motion_avatar_amd.dist import scatter_frames

    tokens_total = total_frames + 1
    src = torch.arange(tokens_total * 6, dtype=torch.float32).view(tokens_total, 2, 3) if rank == 0 else None
    mine = scatter_frames(src, tokens_total, (2, 3), torch.float32, "cpu")
    ts, te = shard_range(tokens_total, world, rank)
    ok = torch.equal(mine, torch.arange(tokens_total * 6, dtype=torch.float32).view(tokens_total, 2, 3)[ts:te])
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    if rank == 0:
        out.put((full[:, 0, 0, 0].tolist(), flags))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_all_gather_reassembles_the_clip_in_order():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    total = 12
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    frames, token_blocks_ok = got
    assert frames == list(range(total))
    assert token_blocks_ok == [True, True]
