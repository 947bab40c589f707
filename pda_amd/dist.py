"""Item-parallel evaluation across the GPUs of one node (the reference is single-device).

One process per GPU.  Rank r owns the item rows [lo_r, hi_r) (+ their popularity); the user table and the
history CSR are replicated.  Per user block every rank produces its partial top-K (packed keys), then ONE
RCCL all-gather over xGMI moves Bu*K*8 bytes per rank and every rank merges the R lists with
pda_topk_merge.  The all-gather + merge of block b runs on a side stream while block b+1 is being scored.

`score_fn` / `merge_fn` default to the HIP entry points; tests inject doubles to exercise the
orchestration under gloo on CPU (there is no CPU product path).
"""
from __future__ import annotations

from typing import Callable, Iterable, Iterator, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous item range of `rank`, sized in whole 32-item MFMA tiles so shards stay balanced."""
    tiles = (n_items + 31) // 32
    per = ((tiles + world - 1) // world) * 32
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def _all_gather_keys(keys: torch.Tensor, world: int, group=None) -> torch.Tensor:
    out = torch.empty((world,) + tuple(keys.shape), dtype=keys.dtype, device=keys.device)
    if dist.get_backend(group) == "gloo":
        parts = list(out.unbind(0))
        dist.all_gather(parts, keys.contiguous(), group=group)
        return torch.stack(parts)
    dist.all_gather_into_tensor(out, keys.contiguous(), group=group)
    return out


class ItemShardedTopK:
    def __init__(self, U: torch.Tensor, I_shard: torch.Tensor, item_offset: int, pop_shard: Optional[torch.Tensor] = None,
                 rank: int = 0, world: int = 1, group=None, score_fn: Optional[Callable] = None,
                 merge_fn: Optional[Callable] = None):
        if score_fn is None or merge_fn is None:
            from . import ops
            score_fn = score_fn or ops.score_topk_keys
            merge_fn = merge_fn or ops.topk_merge
        self.U, self.I_shard, self.pop_shard, self.item_offset = U, I_shard, pop_shard, item_offset
        self.rank, self.world, self.group = rank, world, group
        self.score_fn, self.merge_fn = score_fn, merge_fn
        self._side = torch.cuda.Stream() if U.is_cuda and world > 1 else None

    @classmethod
    def from_full_tables(cls, U, I_full, pop_full=None, rank=0, world=1, **kw) -> "ItemShardedTopK":
        lo, hi = shard_range(I_full.shape[0], rank, world)
        pop = None if pop_full is None else pop_full[lo:hi].contiguous()
        return cls(U, I_full[lo:hi].contiguous(), lo, pop, rank, world, **kw)

    def set_popularity(self, pop_full: Optional[torch.Tensor]):
        """evaluation.set_testing_popularity (MF/train_new_api.py:710): slice the new vector for this shard."""
        n = self.I_shard.shape[0]
        self.pop_shard = None if pop_full is None else pop_full[self.item_offset:self.item_offset + n].contiguous()

    # -- one block, blocking ---------------------------------------------------------------------
    def local_keys(self, users, K, head, hist):
        keys = self.score_fn(self.U, self.I_shard, users, K, head, self.pop_shard if head else None, hist,
                             self.item_offset, 0)
        if self.world == 1:
            return keys
        return self.merge_fn(keys, users, hist, want="keys")          # [Bu, K] packed, this shard only

    def topk(self, users, K=50, head=0, hist=None):
        keys = self.local_keys(users, K, head, hist)
        if self.world > 1:
            keys = _all_gather_keys(keys, self.world, self.group)     # [R, Bu, K] -- the one collective
        return self.merge_fn(keys, users, hist, want="idx_val")

    # -- many blocks, collective + merge of block b overlapped with scoring of block b+1 ----------
    def topk_blocks(self, blocks: Iterable[torch.Tensor], K=50, head=0, hist=None) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        if self.world == 1 or self._side is None:
            for users in blocks:
                yield self.topk(users, K, head, hist)
            return
        main = torch.cuda.current_stream()
        pending = None                       # (result, done-event) of the previous block, produced on the side stream

        def hand_over(p):
            res, done = p
            main.wait_event(done)            # orders only what the consumer enqueues next; scoring of b+1 is already queued
            for t in res:
                t.record_stream(main)
            return res

        for users in blocks:
            keys = self.local_keys(users, K, head, hist)
            ev = torch.cuda.Event()
            ev.record(main)
            if pending is not None:
                yield hand_over(pending)
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                keys.record_stream(self._side)
                allk = _all_gather_keys(keys, self.world, self.group)
                res = self.merge_fn(allk, users, hist, want="idx_val")
                done = torch.cuda.Event()
                done.record(self._side)
                pending = (res, done)
        if pending is not None:
            yield hand_over(pending)
