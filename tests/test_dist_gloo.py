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


def _cache_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from audio_motion_avatar_amd.prediction_cache import PredictionCache

    cache = PredictionCache(cache_replacement_prob=1.0)
    tri = torch.full((1, 6, 4, 12), float(rank + 1))
    smpl = torch.full((1, 6, 4, 2), float(10 * (rank + 1)))
    # every rank works on its own window (batch ids 0 and 1) and publishes the window 12 frames on
    out_tri, out_smpl, used = cache.step(rank, tri[:, :2], smpl[:, :2], lambda a, b: (tri, smpl))
    keys = sorted(cache.entries)
    vals = {k: (float(v["triplane"].mean()), tuple(v["triplane"].shape), v["iter"]) for k, v in cache.entries.items()}
    # a second step on the frame the OTHER rank published: the cached tokens replace the inputs
    other = 1 - rank
    seen = {}
    cache.step(12 + other, tri[:, :2] * 0, smpl[:, :2] * 0, lambda a, b: (seen.setdefault("tri", a), (tri, smpl))[1])
    if rank == 0:
        out.put((keys, vals, float(seen["tri"].mean()), used))
    dist.barrier()
    dist.destroy_process_group()


def test_prediction_cache_entries_reach_every_rank():
    """SURVEY 8(f) row 4 (training-side remainder): lightning_model_wrapper.py:443-493 -- a rank's new cache entry
    (its window's last two outputs, keyed 12 frames ahead) is all-gathered as an object and replaces the encoder's tokens
    when that frame comes up, on whichever rank."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cache_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    keys, vals, replaced_mean, used = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert keys == [(0, 12), (0, 13)] and used == 0
    assert vals[(0, 12)] == (1.0, (1, 2, 4, 12), 1) and vals[(0, 13)] == (2.0, (1, 2, 4, 12), 1)
    assert replaced_mean == 2.0  # rank 0 started frame 13 from rank 1's prediction, not from its zeroed inputs


def test_prediction_cache_rules_without_a_process_group():
    import random

    from audio_motion_avatar_amd.prediction_cache import PredictionCache

    off = PredictionCache(0.0)
    t, s = torch.ones(1, 2, 3, 4), torch.ones(1, 2, 3, 2)
    assert off.store(0, t, s, 0) is None and off.maybe_replace(0, t, s) == (t, s, 0) and not off.entries
    cache = PredictionCache(0.5, rng=random.Random(3))
    item = cache.store(5, torch.arange(6.0).view(1, 6, 1, 1), torch.arange(6.0).view(1, 6, 1, 1), 2)
    (key, entry), = item.items()
    assert key == (0, 17) and entry["iter"] == 3 and entry["triplane"].flatten().tolist() == [4.0, 5.0]
    assert cache.store(5, t, s, PredictionCache.MAX_ITERATIONS) is None      # re-used too often: not extended
    hits = sum(cache.maybe_replace(17, t, s)[2] == 3 for _ in range(400))
    assert 150 < hits < 250                                                  # replaced with probability 0.5
    assert cache.maybe_replace(18, t, s)[2] == 0                             # no entry for that frame
