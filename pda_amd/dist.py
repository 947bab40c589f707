"""Item-parallel evaluation (and SGD training) across the GPUs of one node (the reference is single-device).

One process per GPU.  Rank r owns the item rows [lo_r, hi_r) (+ their popularity); the user table and the
history CSR are replicated.  Per user block every rank produces its partial top-K (packed keys), then ONE RCCL
collective over xGMI and a merge with pda_topk_merge:
    topk           all-gather (Bu*K*8 bytes per rank out, R times that in): every rank ends up with the full [Bu, K]
    topk_sharded   all-to-all (R times less traffic, R-1 links in parallel on the full mesh): rank r merges and keeps the
                   lists of ITS slice of the users -- what bench.py times for N > 1 (at R = 8 the all-gather would move
                   183 MB per rank and 65 536-user block, more time than the scoring)
The collective + merge of block b run on a side stream while block b+1 is being scored.
Early-terminating sweeps add the seed exchange (ops.seeded_begin / seeded_counts / seeded_finish): ONE all-reduce MAX of 12 bytes
per user and, from four shards on, ONE all-reduce SUM of 28 bytes per user -- both depend on the block's warm-up only and are
issued for block b + 1 on a second side stream under the sweep of block b (`topk_blocks`): at most three collectives per block.

From three item shards on, sweeps of the popularity head replicate the 256 globally most popular rows on every rank and take them out
of the shards (round 4, `_topk_blocks_hot`): a rank warms up only its 1 / R of the users on them, ONE all-gather of 4 bytes per user turns
their K-th values into every rank's seed, and the cold shards are swept from empty lists (ops.sweep_from_seed) -- two collectives per
block, and no rank pays an exact warm-up for all users any more.

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


def grid_layout(rank: int, world: int, user_groups: int):
    """Two-dimensional layout of the node: `user_groups` groups of world / user_groups ranks.  Inside a group the catalogue
    is item-sharded (every rank owns a slice, partial lists, one collective among the group's ranks); the groups take
    different users of a block and never talk to each other.  Why: the per-block fixed cost of a rank (the exact warm-up on
    its slice's first 256 items, list hand-over, launches: ~0.9 ms per 262 144 users, independent of the slice size) does not
    shrink with the item shard -- 8 item shards of config 3 give 13.6 / 2.71 = 5.0x --, but it does shrink with the users:
    2 groups x 4 shards 5.9x, 4 groups x 2 shards 6.8x (default_user_groups).  Returns (group index, rank inside the group,
    ranks of the group)."""
    if user_groups < 1 or world % user_groups != 0:
        raise ValueError("world size must be a multiple of the number of user groups")
    per = world // user_groups
    g = rank // per
    return g, rank % per, list(range(g * per, (g + 1) * per))


def default_user_groups(world: int) -> int:
    """1: the catalogue item-sharded over ALL ranks of the job, one list exchange among them -- BASELINE config 4's layout, and since the
    replicated hot items (round 4: `_topk_blocks_hot`) predicted at 6.1 - 6.9 x one GPU at eight ranks from per-rank steps (DESIGN.md section 4;
    unmeasured on eight GPUs).  PDA_USER_GROUPS overrides (grid_user_groups(world) = the two-dimensional layout of rounds 2 - 4)."""
    import os
    forced = os.environ.get("PDA_USER_GROUPS")
    if forced:
        return int(forced)
    return 1


def grid_user_groups(world: int) -> int:
    """world / 2 user groups from four GPUs on: item shards of 2 x as many user groups as are left (grid_layout) -- the layout bench.py
    times BESIDE the default one.  Measured per rank for one 262 144-user step of config 3 on one MI355X (round 4, dense sweep, the exchange
    overlaps): 4 x 2 1.24 ms, 2 x 4 1.41 ms, item shards only 1.25 - 1.40 ms, against 8.55 ms on one GPU."""
    return world // 2 if (world >= 4 and world % 2 == 0) else 1


def make_item_group(rank: int, world: int, user_groups: int):
    """Process groups of the layout above: EVERY rank has to create every group (torch.distributed contract); returns
    (group index, rank in group, group size, this rank's process group or None for a single group spanning the world)."""
    g, r, ranks = grid_layout(rank, world, user_groups)
    if user_groups == 1:
        return 0, rank, world, None
    mine = None
    for gi in range(user_groups):
        _, _, rk = grid_layout(gi * (world // user_groups), world, user_groups)
        pg = dist.new_group(ranks=rk)
        if gi == g:
            mine = pg
    return g, r, len(ranks), mine


def _all_gather_keys(keys: torch.Tensor, world: int, group=None) -> torch.Tensor:
    out = torch.empty((world,) + tuple(keys.shape), dtype=keys.dtype, device=keys.device)
    if dist.get_backend(group) == "gloo":
        parts = list(out.unbind(0))
        dist.all_gather(parts, keys.contiguous(), group=group)
        return torch.stack(parts)
    dist.all_gather_into_tensor(out, keys.contiguous(), group=group)
    return out


def _exchange_user_slices(keys: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """All-to-all of the packed partial lists: keys [Bu, K] (this rank's item shard, all users) -> [R, Bu/R, K], where
    entry r holds rank r's list for MY slice of the users.  Each rank ships (R-1)/R of its keys once and receives as much
    -- R times less than the all-gather -- and on the xGMI full mesh the R-1 transfers run on R-1 separate links."""
    Bu, K = keys.shape
    out = torch.empty((world, Bu // world, K), dtype=keys.dtype, device=keys.device)
    if dist.get_backend(group) == "gloo" and keys.is_cuda:       # plumbing check on one GPU (tests/test_gpu_two_rank.py)
        parts = list(torch.empty((world, Bu, K), dtype=keys.dtype, device=keys.device).unbind(0))
        dist.all_gather(parts, keys.contiguous(), group=group)
        r = dist.get_rank(group)
        per = Bu // world
        return torch.stack([p[r * per:(r + 1) * per] for p in parts])
    dist.all_to_all_single(out.view(-1), keys.contiguous().view(-1), group=group)
    return out


def _ops_score_fn():
    from . import ops
    return ops.score_topk_keys


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
        # early-terminating sweeps exchange K-th values across the shards (local_keys); a caller's own score_fn opts in
        self.seeded = score_fn is _ops_score_fn()
        self.prune = None            # None: ops' default per head; else passed through to score_fn (the same on every rank)
        self._side = torch.cuda.Stream() if U.is_cuda and world > 1 else None
        self._seed_stream = torch.cuda.Stream() if U.is_cuda and world > 1 else None
        # (begin, counts, finish) of the seeded sweep as separate calls: topk_blocks pipelines them across blocks.  Default:
        # ops.seeded_* when score_fn is ops.score_topk_keys; tests inject doubles.
        self.seeded_api = None
        self.n_collectives = 0       # collectives issued by this object (tests: at most three per user block)
        # Replicated hot items (round 4; dense sweeps of the popularity head, `topk_blocks(sharded=True)`): see _hot_state.
        # hot_items = how many of the globally most popular rows live on every rank (0 = off); sweep_seed_fn / kth_fn default to
        # ops.sweep_from_seed / ops.kth_value when score_fn is ops.score_topk_keys (tests inject doubles)
        self.hot_items = 256 if score_fn is _ops_score_fn() else 0
        # from how many item shards on: at two shards a rank's own warm-up costs about what the hot pass and the remaps do (config 3, 65 536
        # users x 100 000 items per rank: 1.24 ms plain, 1.27 - 1.38 ms with hot items; eight shards: 1.85 vs 1.25 - 1.40).  PDA_HOT_ITEMS_MIN_SHARDS
        # overrides (the same on every rank; tests run two ranks through the path)
        import os
        self.hot_min_shards = int(os.environ.get("PDA_HOT_ITEMS_MIN_SHARDS", "3"))
        # ... and at two shards from this many users per block on (half a warm-up saved per rank: 262 144 users x 100 000 items 4.45 vs 4.80 ms,
        # 131 072 users 2.3 - 2.5 vs 2.53 ms; profiles/round4_hot_items.txt)
        self.hot_min_users_two_shards = 131072
        self.sweep_seed_fn = self.kth_fn = None
        self.n_epoch_collectives = 0  # collectives per weight / popularity version (the hot rows), not per user block
        self._hot = None

    @classmethod
    def from_full_tables(cls, U, I_full, pop_full=None, rank=0, world=1, **kw) -> "ItemShardedTopK":
        lo, hi = shard_range(I_full.shape[0], rank, world)
        pop = None if pop_full is None else pop_full[lo:hi].contiguous()
        ev = cls(U, I_full[lo:hi].contiguous(), lo, pop, rank, world, **kw)
        ev._pop_full = pop_full
        return ev

    def set_popularity(self, pop_full: Optional[torch.Tensor]):
        """evaluation.set_testing_popularity (MF/train_new_api.py:710): slice the new vector for this shard.
        The slice OBJECT is kept while the caller's vector is unchanged (same tensor, same version): the visiting order,
        the item prep and the popularity check of ops are cached per tensor object, and recommend_device calls this once
        per user block."""
        n = self.I_shard.shape[0]
        if pop_full is None:
            if self.pop_shard is not None:
                self._pop_epoch = getattr(self, "_pop_epoch", 0) + 1
            self.pop_shard, self._pop_src, self._pop_full = None, None, None
            return
        src = getattr(self, "_pop_src", None)
        if src is not None and src[0]() is pop_full and src[1] == pop_full._version and self.pop_shard is not None:
            return
        if pop_full.numel() < self.item_offset + n:
            raise ValueError("popularity vector has %d entries, this shard needs items up to %d" % (pop_full.numel(), self.item_offset + n))
        self.pop_shard = pop_full[self.item_offset:self.item_offset + n].contiguous()
        self._pop_full = pop_full
        # (what _validate_once keys on: a count of re-slicings, the same on every rank -- object ids are recycled, and a rank that
        # hit a stale id would skip the flag all-reduce the other ranks issue)
        self._pop_epoch = getattr(self, "_pop_epoch", 0) + 1
        import weakref
        self._pop_src = (weakref.ref(pop_full), pop_full._version)

    # -- one block, blocking ---------------------------------------------------------------------
    def _seed_reduce(self, bounds):
        """The shards' bounds of a user's final K-th value, in place: ONE all-reduce MAX over float32 [3, Bu] = (K-th warm-up
        value, ceil(K / R)-th, minus the ceil(K / R)-th: min x = -max -x) -- ops.seeded_begin."""
        self.n_collectives += 1
        dist.all_reduce(bounds, op=dist.ReduceOp.MAX, group=self.group)

    def _seed_sum(self, counts):
        """The shards' counts of warm-up entries at or above the common thresholds, summed: ONE all-reduce SUM over int32
        [n_thr, Bu] (the grid three sequential bisection rounds used to walk) -- ops.seeded_counts."""
        self.n_collectives += 1
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.group)

    def _seed_applies(self, K, head) -> bool:
        from . import ops
        return self.world > 1 and self.seeded and ops.seed_exchange_applies(self.I_shard.shape[1], K, head, self.prune)

    def _validate_once(self, head):
        """Conditions that differ from rank to rank (the shard's size, its popularity slice) are checked on every rank BEFORE the
        first collective of a seeded sweep and the verdict is shared: a rank that raised alone would leave the others waiting in
        an all-reduce forever.  Once per (set_popularity epoch, head)."""
        key = (head, getattr(self, "_pop_epoch", 0))          # rank-invariant: every rank makes the same set_popularity calls
        if getattr(self, "_validated", None) == key:
            return
        err = ""
        if self.I_shard.shape[0] > (1 << 26):
            err = "seeded item-sharded evaluation: at most 2^26 item rows per shard"
        elif head and self.pop_shard is None and self.I_shard.shape[0] > 0:
            err = "the popularity head needs a popularity vector"
        elif head and self.pop_shard is not None and self.pop_shard.numel() and bool((self.pop_shard < 0).any()):
            err = "pop_shard must be >= 0 (it is pop**gamma of a normalised count)"
        flag = torch.tensor([1.0 if err else 0.0], dtype=torch.float32, device=self.U.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        if float(flag[0]) != 0.0:
            raise ValueError(err or "another rank of the item group rejected its shard (see its message)")
        self._validated = key

    def _empty_shard_seed(self, users, K):
        """A rank without items joins the block's seed collectives with neutral values."""
        from . import ops
        nu, dev = users.numel(), users.device
        b = torch.empty((3, nu), dtype=torch.float32, device=dev)
        b[0:2] = float("-inf")
        b[2] = float("inf")
        self._seed_reduce(b)
        n_thr = ops.seed_thresholds(self.world)
        if n_thr > 0:
            self._seed_sum(torch.zeros((n_thr, nu), dtype=torch.int32, device=dev))

    def local_keys(self, users, K, head, hist):
        seeded = self._seed_applies(K, head)
        if seeded:
            self._validate_once(head)
        if self.I_shard.shape[0] == 0:
            # more ranks than 32-item tiles: this rank owns nothing and contributes empty lists (key 0 = empty slot), so that
            # the collectives of the other ranks do not wait for a call that would fail on a 0-row shard
            if seeded:
                self._empty_shard_seed(users, K)
            return torch.zeros((users.numel(), K) if self.world > 1 else (1, users.numel(), K), dtype=torch.int64, device=users.device)
        extra = {} if self.prune is None else {"prune": self.prune}
        # (fewer warm-up tiles per shard pay only together with the seed exchange -- ops.seeded_begin picks 4 / R there: an UNSEEDED
        # shard with one warm tile starts its sweep with thresholds from 64 items and drowns in candidates: 3.47 instead of 2.33 ms
        # per 262 144 users on a 25 000-item shard, profiles/round3_shard_scaling.txt)
        if seeded:
            # early-terminating sweeps: without the exchange every shard prunes against its own shard's K-th value only and
            # eight shards score 6.5 x the tiles of one GPU between them; with it 1.03 x (0.98 x on four, 1.07 x on two)
            extra["seed_reduce"], extra["seed_shards"], extra["seed_sum"] = self._seed_reduce, self.world, self._seed_sum
        keys = self.score_fn(self.U, self.I_shard, users, K, head, self.pop_shard if head else None, hist,
                             self.item_offset, 0, **extra)
        if self.world == 1:
            return keys
        return self.merge_fn(keys, users, hist, want="keys")          # [Bu, K] packed, this shard only

    def topk(self, users, K=50, head=0, hist=None):
        keys = self.local_keys(users, K, head, hist)
        if self.world > 1:
            self.n_collectives += 1
            keys = _all_gather_keys(keys, self.world, self.group)     # [R, Bu, K] -- the one collective
        return self.merge_fn(keys, users, hist, want="idx_val")

    def user_slice(self, n_users: int) -> Tuple[int, int]:
        """Rows of a user block whose final lists THIS rank produces in sharded mode."""
        per = n_users // self.world
        return self.rank * per, (self.rank + 1) * per

    def _finish(self, keys, users, hist, sharded: bool):
        """The exchange + final merge of one block.  sharded: all-to-all, this rank merges its slice of the users and
        returns (idx, val) for those rows only; else all-gather and every rank merges everything."""
        self.n_collectives += 1
        if sharded and users.numel() % self.world == 0 and (hist is None or getattr(hist, "mode", 1) == 1):
            lo, hi = self.user_slice(users.numel())
            allk = _exchange_user_slices(keys, self.world, self.group)
            return self.merge_fn(allk, users[lo:hi].contiguous(), hist, want="idx_val")
        allk = _all_gather_keys(keys, self.world, self.group)
        res = self.merge_fn(allk, users, hist, want="idx_val")
        if sharded:                                   # block not divisible by the world size: slice the full result
            per = -(-users.numel() // self.world)
            lo = min(self.rank * per, users.numel())
            return tuple(t[lo:lo + per] for t in res)
        return res

    def topk_sharded(self, users, K=50, head=0, hist=None):
        """(idx, val) of this rank's user slice (`user_slice`): the multi-GPU path with R times less traffic; the union
        over ranks is what `topk` returns on every rank."""
        keys = self.local_keys(users, K, head, hist)
        if self.world == 1:
            return self.merge_fn(keys, users, hist, want="idx_val")
        return self._finish(keys, users, hist, True)

    # -- replicated hot items: item shards without a warm-up per rank (round 4) -------------------------------------------------
    # Every rank of an item-sharded sweep used to pay the exact warm-up (its shard's 256 most popular items) for ALL users of a step,
    # whatever its share of the catalogue: 0.55 of a 1.85 ms step at eight shards of config 3 -- 4.6 x on eight GPUs.  Now the
    # `hot_items` globally most popular rows live on every rank (one all-reduce of 128 KB per weight version) and are taken OUT of the
    # shards.  Per user block: rank r scores ITS 1 / R of the users against the hot rows (an exact warm-up of Bu / R users), the
    # K-th values are all-gathered (4 bytes per user) as every user's seed, and each rank sweeps its cold shard from EMPTY lists against
    # the seed (ops.sweep_from_seed: pda_score_topk4_phase_*, phase 4).  The all-to-all of the lists follows as before; the owner of
    # a user slice merges the R cold lists and its hot list.  Two collectives per block.  Exact: a pair of the final top K reaches the
    # K-th value of any K items, so it is a hot pair or a cold pair at or above the seed.  Dense sweeps and early-terminating ones (the
    # product default for this head: the cold sweep then stops where a one-GPU sweep would -- its seed is that sweep's warm-up value).
    def _hot_applies(self, K, head, hist, users, sharded) -> bool:
        if not (self.hot_items and self.world >= 2 and sharded and head == 1 and self.prune in ("order", True, None)):
            return False
        if self.world < self.hot_min_shards and users.numel() < self.hot_min_users_two_shards:
            return False
        if getattr(self, "_pop_full", None) is None or users.numel() % self.world != 0:
            return False
        if hist is not None and getattr(hist, "mode", 1) != 1:          # block-row histories: the caller's rows, not user ids
            return False
        fns = self._hot_fns()
        if fns is None or self._pop_full.numel() < 4 * self.hot_items:
            return False
        if self.sweep_seed_fn is not None and self.kth_fn is not None:
            return True                                         # (a caller's own functions: its business)
        # (the library's hot pass, its K-th values and the sweep from a seed are generation 4's: d 64 / 128 / 256, K <= 54 -- the same on every
        # rank, so that other shapes fall back to topk_sharded everywhere at once)
        from . import ops
        d = int(self.U.shape[1])
        return d in (64, 128, 256) and K <= ops.TOPK_K_V4

    def _hot_fns(self):
        if self.sweep_seed_fn is not None and self.kth_fn is not None:
            return self.sweep_seed_fn, self.kth_fn
        if self.score_fn is _ops_score_fn():
            from . import ops
            return ops.sweep_from_seed, ops.kth_value
        return None

    def _hot_state(self):
        """Hot / cold tables of this rank for the current popularity and weights (cached on both)."""
        key = (getattr(self, "_pop_epoch", 0), self.I_shard._version, self.hot_items)
        if self._hot is not None and self._hot["key"] == key:
            return self._hot
        pop_full, lo, n, dev = self._pop_full, self.item_offset, self.I_shard.shape[0], self.I_shard.device
        H = self.hot_items
        # the H most popular items of the WHOLE catalogue, ties to the lower id (the same on every rank), then in id order: local
        # ids of the hot table, of the cold shard and global ids all ascend together -- remapped lists stay sorted
        hot_ids = torch.argsort(pop_full, descending=True, stable=True)[:H].sort().values          # int64 [H]
        mine = (hot_ids >= lo) & (hot_ids < lo + n)
        hot_I = torch.zeros((H, self.I_shard.shape[1]), dtype=self.I_shard.dtype, device=dev)
        if n > 0:
            hot_I[mine] = self.I_shard[hot_ids[mine] - lo]
        self.n_epoch_collectives += 1
        if hot_I.dtype == torch.bfloat16:            # (a sum of one row and zeros: exact in any dtype; gloo has no bf16 sum)
            t = hot_I.float()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            hot_I = t.to(torch.bfloat16)
        else:
            dist.all_reduce(hot_I, op=dist.ReduceOp.SUM, group=self.group)
        cold_mask = torch.ones(n, dtype=torch.bool, device=dev)
        if n > 0:
            cold_mask[hot_ids[mine] - lo] = False
        cold_local = torch.nonzero(cold_mask).flatten()
        st = {"key": key, "hot_ids": hot_ids, "hot_I": hot_I.contiguous(), "hot_pop": pop_full[hot_ids].contiguous(),
              "hot_gid": hot_ids.to(torch.int32),
              "I_cold": self.I_shard[cold_local].contiguous(), "pop_cold": self.pop_shard[cold_local].contiguous(),
              "cold_gid": (cold_local + lo).to(torch.int32), "cold_index_of": torch.cumsum(cold_mask, 0) - 1, "hist": None}
        self._hot = st
        return st

    def _hot_hist(self, st, hist):
        """The caller's history (rows = user ids, GLOBAL item ids ascending) restricted to the hot table and to this rank's cold shard,
        in their local ids (still ascending).  Cached per history object."""
        if hist is None:
            return None, None
        if st["hist"] is not None and st["hist"][0] is hist:
            return st["hist"][1], st["hist"][2]
        as_tuple = isinstance(hist, tuple)
        indptr = torch.as_tensor(hist[0]) if as_tuple else hist.indptr
        indices = torch.as_tensor(hist[1]) if as_tuple else hist.indices
        dev, n_rows = indices.device, indptr.numel() - 1
        hot_ids, lo, n = st["hot_ids"].to(dev), self.item_offset, self.I_shard.shape[0]
        rows = torch.repeat_interleave(torch.arange(n_rows, device=dev), indptr[1:] - indptr[:-1])
        idx = indices.long()
        pos = torch.searchsorted(hot_ids, idx).clamp_(max=hot_ids.numel() - 1)
        is_hot = hot_ids[pos] == idx
        loc = idx - lo
        is_cold = (loc >= 0) & (loc < n) & ~is_hot

        def csr(sel, local_ids):
            ptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
            torch.cumsum(torch.bincount(rows[sel], minlength=n_rows), 0, out=ptr[1:])
            ix = local_ids.to(torch.int32).contiguous()
            if as_tuple:
                return ptr.numpy(), ix.numpy()
            return type(hist)(ptr, ix, by_user=True)

        h_hot = csr(is_hot, pos[is_hot])
        h_cold = csr(is_cold, st["cold_index_of"].to(dev)[loc[is_cold]])
        st["hist"] = (hist, h_hot, h_cold)
        return h_hot, h_cold

    @staticmethod
    def remap_keys(keys: torch.Tensor, gid: torch.Tensor) -> torch.Tensor:
        """Packed keys whose item field holds LOCAL row ids -> the same keys with gid[local] (empty slots stay 0).  gid ascends with
        the local id, so a sorted list stays sorted (ties included)."""
        if keys.is_cuda:                       # one fused launch, in place (the torch expression below is eight passes over the keys)
            from . import ops
            return ops.remap_key_items(keys.contiguous(), gid)
        low = keys & 0xFFFFFFFF
        item = (0xFFFFFFFF - low).clamp_(0, max(0, gid.numel() - 1))
        out = (keys - low) | (0xFFFFFFFF - gid.long()[item])
        return torch.where(keys == 0, keys, out)

    def _hot_start(self, users, K, head, hist):
        """Hot pass of one block: this rank's slice of the users against the replicated hot rows -> (hot keys with global ids
        [Bu / R, K], this slice's seed [Bu / R])."""
        st = self._hot_state()
        h_hot, _ = self._hot_hist(st, hist)
        _, kth = self._hot_fns()
        lo, hi = self.user_slice(users.numel())
        mine = users[lo:hi].contiguous()
        keys = self.score_fn(self.U, st["hot_I"], mine, K, head, st["hot_pop"], h_hot, 0, 1, prune="order")        # [1, Bu / R, K], sorted
        seed = kth(keys, K - 1)                                                                                    # float32 [Bu / R]
        return self.remap_keys(keys[0], st["hot_gid"]), seed

    def _hot_gather_seed(self, seed_slice: torch.Tensor) -> torch.Tensor:
        self.n_collectives += 1
        out = torch.empty((self.world,) + tuple(seed_slice.shape), dtype=seed_slice.dtype, device=seed_slice.device)
        if dist.get_backend(self.group) == "gloo":
            parts = list(out.unbind(0))
            dist.all_gather(parts, seed_slice.contiguous(), group=self.group)
            return torch.stack(parts).reshape(-1)
        dist.all_gather_into_tensor(out, seed_slice.contiguous(), group=self.group)
        return out.reshape(-1)

    def _hot_sweep(self, users, K, head, hist, seed):
        """Cold pass: the whole block against this rank's shard without the hot rows, from empty lists -> keys [Bu, K], global ids."""
        st = self._hot_state()
        _, h_cold = self._hot_hist(st, hist)
        sweep, _ = self._hot_fns()
        if st["I_cold"].shape[0] == 0:
            return torch.zeros((users.numel(), K), dtype=torch.int64, device=users.device)
        keys = sweep(self.U, st["I_cold"], users, K, head, st["pop_cold"], h_cold, 0, seed, prune=(True if self.prune is None else self.prune))
        keys = keys[0] if keys.shape[0] == 1 else self.merge_fn(keys, users, None, want="keys")     # (the item splits of the sweep)
        return self.remap_keys(keys, st["cold_gid"])

    def _hot_finish(self, cold_keys, hot_keys, users, hist):
        """all-to-all of the cold lists, then the owner of the slice merges them with its hot list."""
        self.n_collectives += 1
        lo, hi = self.user_slice(users.numel())
        allk = _exchange_user_slices(cold_keys, self.world, self.group)                  # [R, Bu / R, K]
        allk = torch.cat([allk, hot_keys[None]], 0)
        return self.merge_fn(allk, users[lo:hi].contiguous(), hist, want="idx_val")

    def topk_hot(self, users, K=50, head=1, hist=None):
        """One block through the replicated-hot-items path, blocking: (idx, val) of this rank's user slice."""
        hot_keys, seed_slice = self._hot_start(users, K, head, hist)
        seed = self._hot_gather_seed(seed_slice)
        cold = self._hot_sweep(users, K, head, hist, seed)
        return self._hot_finish(cold, hot_keys, users, hist)

    # -- many blocks, collective + merge of block b overlapped with scoring of block b+1 ----------
    def _seeded_steps(self):
        """(begin, counts, finish) when the seeded sweep can be driven step by step, else None."""
        if self.seeded_api is not None:
            return self.seeded_api
        if self.score_fn is _ops_score_fn():
            from . import ops
            return ops.seeded_begin, ops.seeded_counts, ops.seeded_finish
        return None

    def topk_blocks(self, blocks: Iterable[torch.Tensor], K=50, head=0, hist=None, sharded: bool = False) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        if self.world == 1:
            for users in blocks:
                yield self.topk_sharded(users, K, head, hist) if sharded else self.topk(users, K, head, hist)
            return
        # dense AND early-terminating sweeps of the popularity head through the replicated hot items (the hot rows' K-th value is the seed a
        # one-GPU warm-up would find); the first block decides for the stream of blocks (a rank-invariant decision: block sizes are)
        import itertools
        it = iter(blocks)
        first = next(it, None)
        if first is None:
            return
        blocks = itertools.chain([first], it)
        if self._hot_applies(K, head, hist, first, sharded):
            yield from self._topk_blocks_hot(blocks, K, head, hist)
            return
        # (a rank without items follows the same order of collectives as the others: same pipeline, neutral values)
        steps = self._seeded_steps() if self._seed_applies(K, head) else None
        if self._side is None and steps is None:
            for users in blocks:
                yield self.topk_sharded(users, K, head, hist) if sharded else self.topk(users, K, head, hist)
            return
        cuda = self._side is not None
        main = torch.cuda.current_stream() if cuda else None
        pending = None                       # (result, done-event) of the previous block, produced on the side stream

        def hand_over(p):
            res, done = p
            if cuda:
                main.wait_event(done)        # orders only what the consumer enqueues next; scoring of b+1 is already queued
                for t in res:
                    t.record_stream(main)
            return res

        def exchange(keys, users):
            """collective + final merge of one block on the side stream"""
            if not cuda:
                return self._finish(keys, users, hist, sharded), None
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                keys.record_stream(self._side)
                res = self._finish(keys, users, hist, sharded)
                done = torch.cuda.Event()
                done.record(self._side)
            return res, done

        if steps is None:
            for users in blocks:
                keys = self.local_keys(users, K, head, hist)
                nxt = exchange(keys, users)
                if pending is not None:
                    yield hand_over(pending)
                pending = nxt
            if pending is not None:
                yield hand_over(pending)
            return

        # Seeded early-terminating sweeps, software-pipelined: main stream  W(b+1)  S(b)  W(b+2)  S(b+1) ...  (W = warm-up + bounds,
        # S = pick + sweep + local merge); seed stream  MAX(b+1), counts(b+1), SUM(b+1)  under S(b); side stream  all-to-all + merge.
        begin, counts, finish = steps
        self._validate_once(head)
        popv = self.pop_shard if head else None

        empty = self.I_shard.shape[0] == 0

        def start(users):
            if empty:
                c = None
            else:
                c = begin(self.U, self.I_shard, users, K, head, popv, hist, self.item_offset, 0, self.world)
            if not cuda or empty:
                if empty:
                    self._empty_shard_seed(users, K)
                    return c, users, None
                self._seed_reduce(c.bounds)
                cnt = counts(c)
                if cnt is not None:
                    self._seed_sum(cnt)
                return c, users, None
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(self._seed_stream):
                self._seed_stream.wait_event(ev)
                self._seed_reduce(c.bounds)
                cnt = counts(c)
                if cnt is not None:
                    self._seed_sum(cnt)
                ready = torch.cuda.Event()
                ready.record(self._seed_stream)
            return c, users, ready

        def sweep(st):
            c, users, ready = st
            if ready is not None:
                main.wait_event(ready)
            if c is None:
                keys = torch.zeros((users.numel(), K), dtype=torch.int64, device=users.device)
            else:
                keys = self.merge_fn(finish(c), users, hist, want="keys")
            return exchange(keys, users)

        waiting = None
        for users in blocks:
            st = start(users)
            if waiting is not None:
                nxt = sweep(waiting)
                if pending is not None:
                    yield hand_over(pending)
                pending = nxt
            waiting = st
        if waiting is not None:
            nxt = sweep(waiting)
            if pending is not None:
                yield hand_over(pending)
            pending = nxt
        if pending is not None:
            yield hand_over(pending)


    def _topk_blocks_hot(self, blocks, K, head, hist):
        """topk_blocks(sharded=True) through the replicated-hot-items path, software-pipelined like the seeded sweeps: main stream
        H(b + 1) S(b) H(b + 2) S(b + 1) ... (H = hot pass of this rank's user slice, S = cold sweep + split merge), the all-gather of
        block b + 1's seed on the seed stream under S(b), the all-to-all + final merge on the side stream.  A block the path does not
        take (a size the ranks cannot split evenly, a block-row history) drains the pipeline and goes through topk_sharded."""
        cuda = self._side is not None
        main = torch.cuda.current_stream() if cuda else None
        # (a rank-local failure -- a negative popularity in this rank's slice -- must not leave the others waiting in the first collective:
        # checked on every rank and shared, once per set_popularity epoch, as the seeded path does)
        self._validate_once(head)

        def hand_over(p):
            res, done = p
            if cuda and done is not None:
                main.wait_event(done)
                for t in res:
                    t.record_stream(main)
            return res

        def start(users):
            hot_keys, seed_slice = self._hot_start(users, K, head, hist)
            if not cuda:
                return users, hot_keys, self._hot_gather_seed(seed_slice), None
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(self._seed_stream):
                self._seed_stream.wait_event(ev)
                seed_slice.record_stream(self._seed_stream)
                seed = self._hot_gather_seed(seed_slice)
                ready = torch.cuda.Event()
                ready.record(self._seed_stream)
            return users, hot_keys, seed, ready

        def sweep(st):
            users, hot_keys, seed, ready = st
            if ready is not None:
                main.wait_event(ready)
                seed.record_stream(main)
            cold = self._hot_sweep(users, K, head, hist, seed)
            if not cuda:
                return self._hot_finish(cold, hot_keys, users, hist), None
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                cold.record_stream(self._side)
                hot_keys.record_stream(self._side)
                res = self._hot_finish(cold, hot_keys, users, hist)
                done = torch.cuda.Event()
                done.record(self._side)
            return res, done

        waiting = pending = None
        for users in blocks:
            if not self._hot_applies(K, head, hist, users, True):
                if waiting is not None:
                    nxt = sweep(waiting)
                    if pending is not None:
                        yield hand_over(pending)
                    pending, waiting = nxt, None
                if pending is not None:
                    yield hand_over(pending)
                    pending = None
                yield self.topk_sharded(users, K, head, hist)
                continue
            st = start(users)
            if waiting is not None:
                nxt = sweep(waiting)
                if pending is not None:
                    yield hand_over(pending)
                pending = nxt
            waiting = st
        if waiting is not None:
            nxt = sweep(waiting)
            if pending is not None:
                yield hand_over(pending)
            pending = nxt
        if pending is not None:
            yield hand_over(pending)


class ItemShardedBPR:
    """Item-parallel SGD training step (north_star: "each rank owns an item-embedding slice, BPR negatives sampled
    locally"; SURVEY 8(e) train row).  Rank r holds the item rows [item_offset, item_offset + n_local) and a replica
    of U.  Its sub-batch (B_local triplets, the same on every rank) has positives and negatives inside the slice, so
    both dots are local.  Per step:

        pda_bpr_step_shard_f32   item rows updated in place; user gradients + user ids + this rank's loss share are
                                 written straight into ONE packed exchange buffer [B_local, d + 4]
        all_gather_into_tensor   the only collective: B_local * (d + 4) * 4 bytes per rank (RCCL over xGMI)
        pda_apply_user_grads_f32 every rank applies all R * B_local user gradients -> the replicas of U stay identical

    which equals one fused SGD step of the single-GPU kernel on the concatenated batch (tests/test_gpu_bpr_step.py) -- like that
    step, the item rows take their updates inside the launch that still gathers them (hogwild within a batch: include/pda_hip.h,
    PDA_UPD_SGD_FUSED); optimizer="adam" accumulates first and is exact.
    The exchange is latency-bound (tens of microseconds against a ~9 us step at B=2048): item-parallel training buys
    capacity, not speed -- every BASELINE config fits one MI355X (288 GB), hence bench.py trains on one GPU.

    `step_fn` / `apply_fn` default to the HIP entry points; tests inject doubles under gloo on CPU."""

    def __init__(self, U: torch.Tensor, I_shard: torch.Tensor, item_offset: int, *, regs: float, lr: float, global_batch: int,
                 rank: int = 0, world: int = 1, group=None, step_fn: Optional[Callable] = None, apply_fn: Optional[Callable] = None,
                 optimizer: str = "sgd", sweep_fn: Optional[Callable] = None):
        if step_fn is None or apply_fn is None:
            from . import ops
            step_fn = step_fn or ops.bpr_step_shard
            apply_fn = apply_fn or ops.apply_user_grads
        if optimizer not in ("sgd", "adam"):
            raise NotImplementedError("item-parallel optimizer must be sgd | adam")
        self.U, self.I_shard, self.item_offset = U, I_shard, item_offset
        self.regs, self.lr, self.global_batch = regs, lr, global_batch
        self.rank, self.world, self.group = rank, world, group
        self.step_fn, self.apply_fn = step_fn, apply_fn
        self.optimizer, self._t, self._state = optimizer, 0, None
        if optimizer == "adam":
            # the reference's optimiser (TF-1.14 Adam, dense decay): U and its state are replicated and swept identically on
            # every rank, the item slice and its state are swept by their owner
            if sweep_fn is None:
                from . import ops
                sweep_fn = ops.adam_dense_sweep
            z = torch.zeros_like
            self._state = {"mU": z(U), "vU": z(U), "gU": z(U), "mI": z(I_shard), "vI": z(I_shard), "gI": z(I_shard)}
        self.sweep_fn = sweep_fn

    def local_step(self, users, pos, neg, pos_pop=None, neg_pop=None) -> torch.Tensor:
        """This rank's kernel: updates the local item rows, returns the packed exchange buffer [B_local, d + 4] =
        [g_user (d) | user id bits | loss, mf, reg shares (row 0 only)]; rows are 16-byte aligned."""
        Bl, d = users.numel(), self.U.shape[1]
        if Bl * self.world != self.global_batch:
            raise ValueError("every rank must bring global_batch / world triplets")
        buf = torch.zeros((Bl, d + 4), dtype=torch.float32, device=users.device)
        extra = {"gI_shard": self._state["gI"]} if self.optimizer == "adam" else {}
        self.step_fn(self.U, self.I_shard, self.item_offset, users, pos, neg, pos_pop, neg_pop, regs=self.regs,
                     reg_div=float(self.global_batch), mean_div=float(self.global_batch), lr=self.lr,
                     g_user=buf[:, :d], loss_acc=buf[0, d + 1:d + 4], **extra)
        buf[:, d].view(torch.int32).copy_(users)
        return buf

    def exchange(self, buf: torch.Tensor) -> torch.Tensor:
        """The one collective of a training step: [B_local, d+4] per rank -> [R * B_local, d+4] everywhere."""
        if self.world == 1:
            return buf
        return _all_gather_keys(buf, self.world, self.group).reshape(self.world * buf.shape[0], buf.shape[1])

    def apply(self, allb: torch.Tensor) -> torch.Tensor:
        """Applies every rank's user gradients to the local replica of U; returns the global (loss, mf, reg)."""
        d = self.U.shape[1]
        users_all = allb[:, d].view(torch.int32).contiguous()
        if self.optimizer == "adam":
            from .ops import adam_lr_t
            st = self._state
            self._t += 1
            lr_t = adam_lr_t(self.lr, self._t)
            self.apply_fn(st["gU"], users_all, allb[:, :d], -1.0)          # gU += every rank's user gradients
            self.sweep_fn(self.U, st["mU"], st["vU"], st["gU"], lr_t)       # identical on every rank
            self.sweep_fn(self.I_shard, st["mI"], st["vI"], st["gI"], lr_t) # this rank's slice
        else:
            self.apply_fn(self.U, users_all, allb[:, :d], self.lr)
        Bl = allb.shape[0] // self.world
        return allb.view(self.world, Bl, d + 4)[:, 0, d + 1:d + 4].sum(dim=0)

    def step(self, users, pos, neg, pos_pop=None, neg_pop=None) -> torch.Tensor:
        """One global step; returns the (loss, mf_loss, reg_loss) of the GLOBAL batch as a device tensor (no sync)."""
        return self.apply(self.exchange(self.local_step(users, pos, neg, pos_pop, neg_pop)))


def local_train_csr(train_indptr: torch.Tensor, train_indices: torch.Tensor, lo: int, hi: int, train_slots: Optional[torch.Tensor] = None):
    """The part of the (replicated) train CSR that falls into the item slice [lo, hi): what a rank samples its
    positives from and rejects its negatives against.  Returns (indptr int64 [n_users+1], indices int32 (global ids),
    slots | None, user_pool int32 = users with at least one positive in the slice).  torch plumbing, done once."""
    keep = (train_indices >= lo) & (train_indices < hi)
    csum = torch.zeros(train_indices.numel() + 1, dtype=torch.int64, device=train_indices.device)
    torch.cumsum(keep.to(torch.int64), 0, out=csum[1:])
    indptr = csum[train_indptr]
    indices = train_indices[keep].contiguous()
    slots = None if train_slots is None else train_slots[keep].contiguous()
    pool = torch.nonzero(indptr[1:] > indptr[:-1]).flatten().to(torch.int32)
    return indptr.contiguous(), indices, slots, pool


class ShardSampler:
    """Device sampler of one rank of ItemShardedBPR: users from the rank's pool (distinct inside the sub-batch), positive
    uniform over the user's history INSIDE the slice, negative uniform over the slice minus that history, popularity
    of the positive's time slot (pda_sample_triplets, the semantics of MF/train_new_api.py:366-412 restricted to a
    slice -- the deviation SURVEY 8(e) declares: negatives are uniform over 1/R of the catalogue)."""

    def __init__(self, train_indptr, train_indices, lo: int, hi: int, B_local: int, *, seed: int, rank: int = 0,
                 train_slots=None, pop_matrix=None):
        self.indptr, self.indices, self.slots, self.pool = local_train_csr(train_indptr, train_indices, lo, hi, train_slots)
        if self.pool.numel() < B_local:
            raise ValueError("fewer users with a positive in this item slice than the sub-batch asks for")
        self.lo, self.hi, self.B, self.seed, self.pop_matrix = lo, hi, B_local, seed + 7919 * rank, pop_matrix

    def __call__(self, step: int):
        from . import ops
        return ops.sample_triplets(self.indptr, self.indices, self.B, seed=self.seed, step=step, user_pool=self.pool,
                                   n_pool=self.pool.numel(), train_slots=self.slots, neg_range=(self.lo, self.hi),
                                   pop_matrix=self.pop_matrix)
