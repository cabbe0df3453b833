"""The stage-2 prediction cache of the reference (src/models/lightning_model_wrapper.py:443-493): with probability
`cache_replacement_prob` a window starts from tokens the model itself predicted for that frame earlier (instead of the
encoder's), so it learns to continue its own roll-outs; every rank's new entries reach all ranks through
`all_gather_object`.  Host-side bookkeeping around the hot path: tokens are stored on the CPU, as the reference does.
"""
import random

import torch
import torch.distributed as dist


class PredictionCache:
    MAX_ITERATIONS = 30   # lightning_model_wrapper.py:470: an entry that has been re-used this often is not extended
    FRAME_OFFSET = 12     # :472: the tokens predicted in window `batch_id` seed the window at `batch_id + 12`

    def __init__(self, cache_replacement_prob=0.0, rng=None):
        self.cache_replacement_prob = float(cache_replacement_prob)
        self.entries = {}  # (subject_id, frame_id) -> {"triplane", "smplx_tokens", "iter"}
        self.rng = rng or random

    def maybe_replace(self, batch_id, triplanes, smplx_tokens, subject_id=0):
        """:443-458 -> (triplanes, smplx_tokens, use_cache): the cached tokens for this frame with probability
        cache_replacement_prob when an entry exists, else the inputs (use_cache = how often the entry was re-used)."""
        if self.cache_replacement_prob > 0 and self.rng.random() < self.cache_replacement_prob:
            hit = self.entries.get((subject_id, batch_id))
            if hit is not None and hit["triplane"] is not None and hit["smplx_tokens"] is not None:
                return hit["triplane"].to(triplanes.device), hit["smplx_tokens"].to(smplx_tokens.device), hit["iter"]
        return triplanes, smplx_tokens, 0

    def store(self, batch_id, output_triplane_tokens, output_smplx_tokens, use_cache, subject_id=0):
        """:469-481 -> the new item {key: entry} (or None): the window's last two outputs, on the CPU, for the window
        FRAME_OFFSET later."""
        if not (self.cache_replacement_prob > 0 and use_cache < self.MAX_ITERATIONS):
            return None
        key = (subject_id, batch_id + self.FRAME_OFFSET)
        entry = {"triplane": output_triplane_tokens[:, -2:].clone().detach().cpu(),
                 "smplx_tokens": output_smplx_tokens[:, -2:].clone().detach().cpu(), "iter": use_cache + 1}
        self.entries[key] = entry
        return {key: entry}

    def sync(self, new_item, group=None):
        """:483-493: every rank's new item reaches every rank (object all-gather; a rank without one sends {})."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        gathered = [None] * dist.get_world_size(group)
        dist.all_gather_object(gathered, new_item or {}, group=group)
        for item in gathered:
            if item:
                self.entries.update(item)

    def step(self, batch_id, triplanes, smplx_tokens, run_window, group=None):
        """One training-style window: replace -> run_window(triplanes, smplx_tokens) -> (tri, smpl) -> store -> sync."""
        triplanes, smplx_tokens, used = self.maybe_replace(batch_id, triplanes, smplx_tokens)
        out_tri, out_smpl = run_window(triplanes, smplx_tokens)
        self.sync(self.store(batch_id, out_tri, out_smpl, used), group)
        return out_tri, out_smpl, used
