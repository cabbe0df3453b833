"""Multi-GPU layer of the rendering path: frame sharding + the one exchange step (all-gather of rendered frames).

One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm (xGMI inside a node).  Frames of a clip are
independent from SMPL-X decode to rasterisation (the reference renders them in a per-frame loop,
src/models/renderer.py:475-477), so each rank owns a contiguous block of frames and nothing is exchanged until the
rendered sequence is reassembled.  On the wire the frames are uint8 RGB (the format the reference writes to video,
src/main2.py:351): 786 KB per 512x512 frame instead of 3.1 MB of fp32 -- and by default only their non-background
16x16 tiles (lossless; see FrameAllGather), because at ~200 k frames/s per GPU the dense frames alone would need more
than the seven xGMI links of a GPU deliver.

The collective is issued on a side stream, double-buffered, so step k's gather overlaps step k+1's rendering.

The autoregressive token generator does not shard in time.  Two modes: "segment" -- every rank
rolls its own chain from the same reference tokens and nothing but frames is exchanged; "sequential" -- one rank runs
the exact chain of the demo loop and hands each rank its block of tokens (send_frames / recv_frames / scatter_frames,
3.2 MB per frame), so only the rendering scales.
"""
import torch
import torch.distributed as dist


def shard_range(total_frames: int, world_size: int, rank: int):
    """Contiguous block of frames owned by `rank` (the first `total % world` ranks take one extra)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, extra = divmod(total_frames, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_gather_frames(local_frames: torch.Tensor, group=None) -> torch.Tensor:
    """[F_local, ...] on every rank (equal F_local) -> [world * F_local, ...] in rank order, on every rank."""
    world = dist.get_world_size(group)
    local_frames = local_frames.contiguous()
    out = torch.empty((world * local_frames.shape[0],) + tuple(local_frames.shape[1:]), dtype=local_frames.dtype,
                      device=local_frames.device)
    dist.all_gather_into_tensor(out, local_frames, group=group)
    return out


def send_frames(block: torch.Tensor, dst: int, group=None):
    """Non-blocking point-to-point send of a contiguous block of per-frame tensors (tokens) to its owner.  RCCL sends
    are ordered on the stream that produced the block; a host-side backend (gloo, used to rehearse the rank logic) reads
    the memory from the CPU, so the producing stream is drained first."""
    block = block.contiguous()
    if block.is_cuda and dist.get_backend(group) != "nccl":
        torch.cuda.current_stream(block.device).synchronize()
    return dist.isend(block, dst=dst, group=group)


def recv_frames(shape, dtype, device, src: int = 0, group=None) -> torch.Tensor:
    """Blocking receive of this rank's block (the counterpart of send_frames)."""
    buf = torch.empty(shape, dtype=dtype, device=device)
    dist.recv(buf, src=src, group=group)
    return buf


def scatter_frames(full, total_frames: int, trailing_shape, dtype, device, src: int = 0, group=None) -> torch.Tensor:
    """Rank `src` holds `full` [total_frames, *trailing_shape]; every rank returns its shard_range block.  Point-to-
    point (the blocks may differ in length by one frame, which scatter() does not allow); xGMI is a full mesh, so the
    sends leave `src` on distinct links."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if rank == src:
        if tuple(full.shape) != (total_frames, *trailing_shape):
            raise ValueError(f"scatter_frames: source holds {tuple(full.shape)}, expected {(total_frames, *trailing_shape)}")
        pending = []
        for dst in range(world):
            s, e = shard_range(total_frames, world, dst)
            if dst != src:
                pending.append(send_frames(full[s:e], dst, group))
        s, e = shard_range(total_frames, world, src)
        mine = full[s:e].clone()
        for req in pending:
            req.wait()
        return mine
    s, e = shard_range(total_frames, world, rank)
    return recv_frames((e - s, *trailing_shape), dtype, device, src, group)


class FrameAllGather:
    """Pack (HIP) + all-gather (RCCL) of a shard's frames on a side stream, two buffers deep.

    wire="sparse" (default): only the 16x16 tiles that differ from the background travel (ops.frames_pack_tiles), and
    every rank unpacks the gathered buffers back into dense uint8 frames -- lossless, ~1/5 of the bytes for an avatar
    clip.  The two dense output buffers are reused, so the unpack is differential: it writes the stored tiles and
    re-clears only the tiles that held pixels in the buffer's previous use (ops.frames_unpack_tiles(state=...)).  The per-rank tile capacity is fixed by calibrate() (a synchronising call, made once before the timed
    region: max stored tiles over the ranks + 10 %); a later step that needs more sets `overflowed()`, exactly like
    the rasterizer's instance capacity.  wire="dense": plain uint8 RGB frames (ops.frames_to_rgb8).
    """

    def __init__(self, frames, height, width, world_size, device, group=None, wire="sparse", bg=(1.0, 1.0, 1.0),
                 algorithm="collective"):
        if wire not in ("sparse", "dense"):
            raise ValueError(f"wire must be 'sparse' or 'dense', got {wire!r}")
        if algorithm not in ("collective", "direct"):
            raise ValueError(f"algorithm must be 'collective' or 'direct', got {algorithm!r}")
        self.algorithm = algorithm
        self.group = group
        self.world = world_size
        self.frames, self.height, self.width = frames, height, width
        self.device = device
        self.wire = wire
        self.bg = tuple(float(c) for c in bg)
        self.stream = torch.cuda.Stream(device=device)
        self.full = [torch.empty(world_size * frames, height, width, 3, dtype=torch.uint8, device=device)
                     for _ in range(2)]
        self.turn = 0
        self._done = [torch.cuda.Event() for _ in range(2)]  # side stream has finished sending local[i]
        self._sent = [False, False]
        if wire == "dense":
            self.local = [torch.empty(frames, height, width, 3, dtype=torch.uint8, device=device) for _ in range(2)]
        else:
            self.capacity = None
            self.status = torch.zeros(1, dtype=torch.int32, device=device)
            # differential unpack: each dense buffer is reused every other step and ~80 % of an avatar frame is
            # background, so only stored tiles and tiles the body moved out of are rewritten.  Frames outside that
            # kernel's limits (width not a multiple of 16, or more tiles per frame than its LDS table holds, e.g.
            # 3840 x 2160) take the full unpack
            from . import ops

            self.tile_state = ([ops.frames_tile_state(world_size, frames, height, width, device) for _ in range(2)]
                               if ops.frames_delta_unpack_supported(height, width) else [None, None])

    # ---- sparse wire -------------------------------------------------------------------------------------------------
    def calibrate(self, rgba: torch.Tensor, headroom=1.1, tile_hint=None):
        """Size the per-rank wire buffers from one rendered shard (host sync + a MAX all-reduce: call it in warm-up).
        Pass the same kind of `tile_hint` the steps will pass to submit(): a hint stores a superset of the tiles."""
        from . import ops

        if self.wire != "sparse":
            return None
        F, H, W = self.frames, self.height, self.width
        count, _ = ops.frames_wire_count(ops.frames_pack_tiles(rgba.view(F, H, W, 4), 0, self.bg, tile_hint=tile_hint))
        t = torch.tensor([count], dtype=torch.int64, device=self.device)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        tiles = F * ((H + 15) // 16) * ((W + 15) // 16)
        self.capacity = min(tiles, int(int(t.item()) * headroom) + 64)
        nbytes = ops.frames_wire_bytes(F, H, W, self.capacity)
        self.local = [torch.empty(nbytes, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.gathered = [torch.empty(self.world, nbytes, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.status.zero_()
        return self.capacity

    def wire_bytes_per_rank(self):
        return int(self.local[0].numel())

    def overflowed(self) -> bool:
        """True when some step since calibrate() had more non-background tiles than the wire holds (synchronises)."""
        return self.wire == "sparse" and bool(int(self.status.item()))

    def _gather(self, out_rows: torch.Tensor, local: torch.Tensor):
        """out_rows [world, ...] <- every rank's `local`, enqueued on the current (side) stream.
        "collective": RCCL's all-gather (its ring / tree choice).  "direct": one grouped send + receive per peer, i.e.
        every rank writes its shard to its 7 peers over 7 distinct xGMI links at once -- the mesh form SURVEY.md
        section 8(e) asks for; the bytes on each link are the same as in the best ring, without the 7 serial hops."""
        if self.algorithm == "collective" or self.world == 1:
            dist.all_gather_into_tensor(out_rows.view(-1), local.view(-1), group=self.group)
            return
        rank = dist.get_rank(self.group)
        if local.is_cuda and dist.get_backend(self.group) != "nccl":
            torch.cuda.current_stream(local.device).synchronize()  # host-side backend: see send_frames()
        peers = [(rank + k) % self.world for k in range(1, self.world)]  # staggered: no two ranks start on one target
        ops_ = []
        for peer in peers:
            ops_.append(dist.P2POp(dist.isend, local, peer, self.group))
            ops_.append(dist.P2POp(dist.irecv, out_rows[peer], peer, self.group))
        reqs = dist.batch_isend_irecv(ops_)
        out_rows[rank].copy_(local, non_blocking=True)
        for req in reqs:
            req.wait()  # RCCL: orders the current stream after the transfers (no host block); gloo: blocks

    # ---- one step ------------------------------------------------------------------------------------------------------
    def submit(self, rgba: torch.Tensor, tile_hint=None, packed=False) -> torch.Tensor:
        """rgba: contiguous fp32 [..., H, W, 4] produced on the current stream.  Returns the (future) full sequence
        [world * F, H, W, 3] uint8; call wait() before reading it on the current stream.  `tile_hint` (sparse wire):
        int32 [F * tiles], zero where the tile is known to be background (RasterWorkspace.tile_counts()), which saves
        one pass over the fp32 frames.  `packed`: the rasterizer has already written wire_target() on the current
        stream: no pack pass at all."""
        from . import ops

        if self.wire == "sparse" and self.capacity is None:
            raise RuntimeError("FrameAllGather(wire='sparse'): call calibrate() once before submit()")
        i = self.turn
        self.turn ^= 1
        F, H, W = self.frames, self.height, self.width
        self.stream.wait_stream(torch.cuda.current_stream())
        rgba.record_stream(self.stream)
        if tile_hint is not None:
            tile_hint.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            if self.wire == "dense":
                ops.frames_to_rgb8(rgba.view(F, H, W, 4), out=self.local[i])
                self._gather(self.full[i].view(self.world, F, H, W, 3), self.local[i])
            else:
                if not packed:
                    ops.frames_pack_tiles(rgba.view(F, H, W, 4), self.capacity, self.bg, wire=self.local[i],
                                          tile_hint=tile_hint)
                self._gather(self.gathered[i], self.local[i])
                ops.frames_unpack_tiles(self.gathered[i], self.world, F, H, W, self.capacity, out=self.full[i],
                                        status=self.status, state=self.tile_state[i])
                self._done[i].record(self.stream)
                self._sent[i] = True
        return self.full[i]

    def wire_target(self):
        """(buffer, capacity) of the wire buffer the NEXT submit() will send: hand it to the rasterizer
        (Renderer.render_tokens(wire=...) / ops.rasterize(wire=...)), which then writes the exchange format itself, and
        call submit(rgba, packed=True) -- the pack pass over the fp32 frames disappears.  Sparse wire, after calibrate()."""
        if self.wire != "sparse" or self.capacity is None:
            raise RuntimeError("wire_target(): sparse wire only, after calibrate()")
        if self._sent[self.turn]:  # the exchange that last used this buffer (two steps ago) must have finished with it
            torch.cuda.current_stream().wait_event(self._done[self.turn])
        return self.local[self.turn], self.capacity

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.stream)
