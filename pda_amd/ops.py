"""Thin, typed front-ends over the C ABI (include/pda_hip.h).  Tensors in, tensors out; every call is
stream-ordered on torch's current HIP stream.  No arithmetic happens here."""
from __future__ import annotations

import math
import os
from typing import Optional, Tuple

import ctypes as C

import torch

from . import _lib
from ._lib import (HEAD_POP, HEAD_RAW, HIST_BY_BLOCK_ROW, HIST_BY_USER_ID, UPD_ANY_ORDER, UPD_DENSE_GRAD, UPD_NONE, UPD_USERS_DISTINCT,  # noqa: F401
                   UPD_SGD_FUSED, check, ptr, stream_ptr)

ADAM_BETA1, ADAM_BETA2, ADAM_EPS = 0.9, 0.999, 1e-8   # tf.train.AdamOptimizer defaults (MF/model_api.py:83)


def _need(t: Optional[torch.Tensor], dtype, name: str, optional=False):
    if t is None:
        if optional:
            return None
        raise ValueError(f"{name} is required")
    if not t.is_cuda:
        raise ValueError(f"{name} must live in HBM (cuda tensor); pda_amd has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous (row-major)")
    return t


class HistoryCSR:
    """Train-item mask in the layout the kernel wants: CSR, int64 indptr, int32 GLOBAL item ids sorted
    ascending inside each row.  ``by_user`` tells whether rows are user ids or block rows
    (the reference builds block-row COO per 2048-user block, MF/train_new_api.py:730-739)."""

    def __init__(self, indptr: torch.Tensor, indices: torch.Tensor, by_user: bool):
        self.indptr = _need(indptr, torch.int64, "hist indptr")
        self.indices = _need(indices, torch.int32, "hist indices")
        self.mode = HIST_BY_USER_ID if by_user else HIST_BY_BLOCK_ROW

    @staticmethod
    def from_lists(rows, device, by_user: bool) -> "HistoryCSR":
        """rows: iterable of per-row item lists (any order, duplicates allowed) -> sorted CSR on device."""
        import numpy as np
        lens = np.fromiter((len(r) for r in rows), dtype=np.int64)
        indptr = np.zeros(len(lens) + 1, dtype=np.int64)
        np.cumsum(lens, out=indptr[1:])
        flat = np.empty(int(indptr[-1]), dtype=np.int32)
        for r, items in enumerate(rows):
            flat[indptr[r]:indptr[r + 1]] = np.sort(np.asarray(items, dtype=np.int32))
        return HistoryCSR(torch.from_numpy(indptr).to(device), torch.from_numpy(flat).to(device), by_user)

    @staticmethod
    def from_coo(index, n_rows: int, device) -> "HistoryCSR":
        """The reference's sparse mask triple index int64[nnz,2] = (row, item) (MF/train_new_api.py:736), built per 2 048-user
        block and handed over with every do_recommendation call (:791): one host -> device copy, then sorted and counted ON THE
        DEVICE (one 64-bit sort of row << 32 | item, a bincount, a cumsum) -- a numpy lexsort of the ~100 000 entries of a block
        cost more than the block's sweep."""
        import numpy as np
        idx = torch.from_numpy(np.ascontiguousarray(np.asarray(index, dtype=np.int64).reshape(-1, 2))).to(device, non_blocking=True)
        if idx.shape[0] == 0:
            return HistoryCSR(torch.zeros(n_rows + 1, dtype=torch.int64, device=device), torch.zeros(0, dtype=torch.int32, device=device), by_user=False)
        # (rows beyond n_rows or negative entries are a malformed mask: an error, as np.add.at / the reference's sparse tensor raise -- one min / max on
        # the device, checked behind the sort)
        bad = (idx[:, 0] < 0) | (idx[:, 0] >= n_rows) | (idx[:, 1] < 0)
        key = torch.sort((idx[:, 0] << 32) | idx[:, 1]).values
        if bool(bad.any()):
            raise IndexError("mask triple: row index outside [0, %d) or a negative item id" % n_rows)
        items = (key & 0xFFFFFFFF).to(torch.int32)
        indptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=device)
        torch.cumsum(torch.bincount(key >> 32, minlength=n_rows)[:n_rows], 0, out=indptr[1:])
        return HistoryCSR(indptr, items, by_user=False)


def auto_splits(n_users_blk: int, n_items_local: int) -> int:
    return _lib.load().pda_score_topk_auto_splits(n_users_blk, n_items_local)


import weakref

# id(tensor) -> (weakref(tensor), tensor._version, prep buffer).  Validated by OBJECT identity, never by address:
# the caching allocator hands the same address to the next table of the same shape.
_PREP_CACHE = {}


def item_prep(I_shard: torch.Tensor) -> torch.Tensor:
    """pda_item_prep_f32 (bf16 hi/lo planes + padded norms of an item shard), cached per weight version: any
    in-place update of the table bumps tensor._version and triggers a re-split."""
    lib = _lib.load()
    key = id(I_shard)
    hit = _PREP_CACHE.get(key)
    if hit is not None and hit[0]() is I_shard:
        if hit[1] == I_shard._version:
            return hit[2]
        buf = hit[2]
    else:
        n, d = I_shard.shape
        nbytes = lib.pda_item_prep_bf16_bytes(n, d) if I_shard.dtype == torch.bfloat16 else lib.pda_item_prep_bytes(n, d)
        buf = torch.empty(nbytes, dtype=torch.uint8, device=I_shard.device)
    if I_shard.dtype == torch.bfloat16:
        check(lib.pda_item_prep_bf16(ptr(I_shard), I_shard.shape[0], I_shard.shape[1], ptr(buf), stream_ptr()), "pda_item_prep_bf16")
    else:
        check(lib.pda_item_prep_f32(ptr(I_shard), I_shard.shape[0], I_shard.shape[1], ptr(buf), stream_ptr()), "pda_item_prep_f32")
    for k in [k for k, v in _PREP_CACHE.items() if v[0]() is None]:
        del _PREP_CACHE[k]                                   # drop entries whose table died
    _PREP_CACHE[key] = (weakref.ref(I_shard), I_shard._version, buf)
    return buf


_ORDER_CACHE = {}      # id(pop or table) -> (weakref, _version, order i32[n])
_PREP_ORD_CACHE = {}   # id(I_shard) -> (weakref(I), I._version, order, pop weakref|None, pop _version, prep buffer)


def visiting_order(I_shard: torch.Tensor, pop_shard: Optional[torch.Tensor]) -> torch.Tensor:
    """The order in which the ordered sweep (pda_score_topk_ordered_f32) visits the shard: most popular first
    (PDA head; does not depend on the weights, so the reordered history survives training), largest norm first (raw
    head).  Only a heuristic -- any permutation gives the same result -- hence computed here, with torch."""
    src = pop_shard if pop_shard is not None else I_shard
    hit = _ORDER_CACHE.get(id(src))
    if hit is not None and hit[0]() is src and hit[1] == src._version:
        return hit[2]
    key = pop_shard.abs() if pop_shard is not None else torch.linalg.vector_norm(I_shard.float(), dim=1)
    order = torch.argsort(key, descending=True, stable=True).to(torch.int32)
    for k in [k for k, v in _ORDER_CACHE.items() if v[0]() is None]:
        del _ORDER_CACHE[k]
    _ORDER_CACHE[id(src)] = (weakref.ref(src), src._version, order)
    return order


def item_prep_ordered(I_shard: torch.Tensor, pop_shard: Optional[torch.Tensor], order: Optional[torch.Tensor] = None):
    """pda_item_prep_ordered_f32, cached per (weight version, pop version, order).  Returns (prep buffer, order)."""
    lib = _lib.load()
    if order is None:
        order = visiting_order(I_shard, pop_shard)
    n, d = I_shard.shape
    hit = _PREP_ORD_CACHE.get(id(I_shard))
    buf = None
    if hit is not None and hit[0]() is I_shard:
        buf = hit[5]
        same_pop = (hit[3] is None and pop_shard is None) or (hit[3] is not None and pop_shard is not None and hit[3]() is pop_shard
                                                             and hit[4] == pop_shard._version)
        if hit[1] == I_shard._version and hit[2] is order and same_pop:
            return buf, order
    bf = I_shard.dtype == torch.bfloat16
    if buf is None:
        nbytes = lib.pda_item_prep_ordered_bf16_bytes(n, d) if bf else lib.pda_item_prep_ordered_bytes(n, d)
        buf = torch.empty(nbytes, dtype=torch.uint8, device=I_shard.device)
    order = _need(order, torch.int32, "order")
    if order.numel() != n:
        raise ValueError("order must have one entry per local item row")
    fn = lib.pda_item_prep_ordered_bf16 if bf else lib.pda_item_prep_ordered_f32
    check(fn(ptr(I_shard), ptr(pop_shard), ptr(order), n, d, ptr(buf), stream_ptr()), "pda_item_prep_ordered")
    for k in [k for k, v in _PREP_ORD_CACHE.items() if v[0]() is None]:
        del _PREP_ORD_CACHE[k]
    _PREP_ORD_CACHE[id(I_shard)] = (weakref.ref(I_shard), I_shard._version, order,
                                    weakref.ref(pop_shard) if pop_shard is not None else None,
                                    pop_shard._version if pop_shard is not None else 0, buf)
    return buf, order


_PREP4_CACHE = {}      # id(I_shard) -> (weakref(I), I._version, order|None, pop weakref|None, pop _version, prep buffer)
SWEEP_WARM_PER_SPLIT = 1024   # PDA_SWEEP_WARM_PER_SPLIT (include/pda_hip.h)
TOPK_K_V4 = 54         # pda_score_topk4_*: K <= 54 (57 list slots per user)


def item_prep4(I_shard: torch.Tensor, pop_shard: Optional[torch.Tensor], order: Optional[torch.Tensor]) -> torch.Tensor:
    """pda_item_prep4_f32 / _bf16 (padded item rows in visiting order + suffix bounds), cached per (weight version, pop
    version, order object).  order None = natural item order."""
    lib = _lib.load()
    n, d = I_shard.shape
    hit = _PREP4_CACHE.get(id(I_shard))
    buf = None
    if hit is not None and hit[0]() is I_shard:
        buf = hit[5]
        same_pop = (hit[3] is None and pop_shard is None) or (hit[3] is not None and pop_shard is not None and hit[3]() is pop_shard
                                                             and hit[4] == pop_shard._version)
        if hit[1] == I_shard._version and hit[2] is order and same_pop:
            return buf
    if buf is None:
        buf = torch.empty(lib.pda_item_prep4_bytes(n, d), dtype=torch.uint8, device=I_shard.device)
    if order is not None:
        order = _need(order, torch.int32, "order")
        if order.numel() != n:
            raise ValueError("order must have one entry per local item row")
    fn = lib.pda_item_prep4_bf16 if I_shard.dtype == torch.bfloat16 else lib.pda_item_prep4_f32
    check(fn(ptr(I_shard), ptr(pop_shard), ptr(order), n, d, ptr(buf), stream_ptr()), "pda_item_prep4")
    for k in [k for k, v in _PREP4_CACHE.items() if v[0]() is None]:
        del _PREP4_CACHE[k]
    _PREP4_CACHE[id(I_shard)] = (weakref.ref(I_shard), I_shard._version, order,
                                 weakref.ref(pop_shard) if pop_shard is not None else None,
                                 pop_shard._version if pop_shard is not None else 0, buf)
    return buf


_PREP7_CACHE = {}


def item_prep7(I_shard: torch.Tensor, order: torch.Tensor) -> torch.Tensor:
    """pda_item_prep7_f32 / _bf16: the funnel's item prep (no popularity, the half-tile image in fp16), cached per (weight version, order object) beside
    the generation-4 prep of the same table (an evaluation epoch alternates between the two heads)."""
    lib = _lib.load()
    n, d = I_shard.shape
    hit = _PREP7_CACHE.get(id(I_shard))
    buf = None
    if hit is not None and hit[0]() is I_shard:
        buf = hit[3]
        if hit[1] == I_shard._version and hit[2] is order:
            return buf
    if buf is None:
        buf = torch.empty(lib.pda_item_prep4_bytes(n, d), dtype=torch.uint8, device=I_shard.device)
    order = _need(order, torch.int32, "order")
    if order.numel() != n:
        raise ValueError("order must have one entry per local item row")
    fn = lib.pda_item_prep7_bf16 if I_shard.dtype == torch.bfloat16 else lib.pda_item_prep7_f32
    check(fn(ptr(I_shard), ptr(order), n, d, ptr(buf), stream_ptr()), "pda_item_prep7")
    for k in [k for k, v in _PREP7_CACHE.items() if v[0]() is None]:
        del _PREP7_CACHE[k]
    _PREP7_CACHE[id(I_shard)] = (weakref.ref(I_shard), I_shard._version, order, buf)
    return buf


def score_kernel(d: int, K: int, nloc: int, prune=None, head: int = HEAD_POP) -> str:
    """Which pre-filtered kernel generation serves a call.  All of them return the same keys; the choice is by measured
    speed (65 536 users per block).  Generation 4 (pda_score_topk_v4.hip: two MFMA waves per SIMD, loader and rescoring
    waves, register-resident exact warm-up) for every sweep in visiting order -- C3 dense 3.40 vs 4.61 ms, early-terminating
    0.42 vs 0.49 ms; C1/C2 0.31 vs 0.60 ms; a config-5 shard 3.2 vs 4.5 ms -- and, since its many-candidates geometry (round 3),
    for the natural-order sweeps at d <= 128; generation 3 for natural order and the raw head at d = 256, whose visiting order
    by norm leaves 300+ candidates per user (19.6 vs 29.3 ms: generation 4 keeps the d = 256 lists in HBM).  PDA_SCORE_KERNEL=v3|v4
    forces one (A/B measurements, cross-checks)."""
    import os
    forced = os.environ.get("PDA_SCORE_KERNEL", "")
    fits = d in (64, 128, 256) and K <= TOPK_K_V4 and nloc <= (1 << 26)
    if forced in ("v3", "old"):
        return "v3"
    if forced == "v4":
        return "v4" if fits else "v3"
    if not fits or (d == 256 and (head == HEAD_RAW or not prune)):
        return "v3"
    return "v4"        # (natural order at d <= 128 too since the many-candidates geometry: C3 65 536 users 7.0 vs 8.8 ms)


# The huge geometry (pda_v5_sweep.h) wants a workgroup of 1 024 users on every CU: a block of fewer than 256 x 1 024 users fills the chip
# with ITEM SPLITS (huge_splits below) -- cheap since round 4's shared warm-up (one exact warm-up per user, not per split)
HUGE_MIN_USERS = 196609         # (kept for callers that pass their own n_splits: from here on the hint is given whatever the split count)
HUGE_MIN_WORKGROUPS = 128       # of 256 CUs
HUGE_SPLIT_MIN_USERS = 4096     # below: the 256-user geometry
HUGE_MIN_TILES_PER_SPLIT = 32   # 64-item tiles: a workgroup's prologue (1 024 thresholds, 256 AGPRs per wave) wants a sweep behind it


def huge_splits(n_users: int, n_items_local: int, d: int = 128) -> int:
    """Item splits with which the huge geometry runs a block of n_users (one workgroup = 1 024 users x one split), or 0 when it should
    not (then the 256-user / wide geometry serves the call).  The time of a launch follows rounds of 256 workgroups x tiles per split:
    cost(S) = ceil(user tiles x S / 256) x (fixed cost of a workgroup + 1 / S), plus a little per split (the merge, the empty splits'
    output rows); the smallest S at the minimum wins.  Measured, config 3, with the shared warm-up (profiles/round4_shared_warm.txt): 8 192 users x 32 splits 0.47 ms
    (256-user geometry 0.58), 16 384 x 16 0.72 (1.01), 32 768 x 8 1.21 (1.82), 65 536 x 4 2.30 (wide 3.1), 131 072 x 2 4.27 (5.6);
    config 2 / config 1 (313 / 407 tiles) x 5 splits 0.37 / 0.39 ms (0.46 / 0.50)."""
    forced = os.environ.get("PDA_HUGE_SPLITS")          # A/B measurements only
    if forced and n_users >= HUGE_SPLIT_MIN_USERS:
        return int(forced)
    return _lib.load().pda_score_topk_huge_splits(n_users, n_items_local, d)      # (the rule lives in the library since round 5: pda_score_topk_plan)


SWEEP_MODE = {None: -1, False: 0, True: 1, "order": 2}
PATH_NAMES = {1: "v1", 2: "v3", 3: "v3", 4: "v4", 7: "funnel"}


def score_plan(n_users: int, n_items_local: int, d: int, K: int = 50, head: int = HEAD_POP, prune=None, bf16: bool = False,
               hist: Optional["HistoryCSR"] = None) -> dict:
    """pda_score_topk_plan: which entry point, item splits, geometry hint, visiting order and workspace the LIBRARY chooses for a call
    (prune: None = its default for the head, False = natural order, True = early-terminating, "order" = dense in visiting order)."""
    p = _lib.ScorePlan()
    check(_lib.load().pda_score_topk_plan(n_users, n_items_local, d, K, head, SWEEP_MODE[prune], int(bool(bf16)),
                                          -1 if hist is None else hist.mode, C.byref(p)), "pda_score_topk_plan")
    return {"path": p.path, "kernel": PATH_NAMES[p.path], "sweep_mode": {0: False, 1: True, 2: "order"}[p.sweep_mode], "n_splits": p.n_splits,
            "early_stop": p.early_stop, "order": p.order, "prep_with_pop": bool(p.prep_with_pop), "workspace_bytes": int(p.workspace_bytes),
            "keys_rows": int(p.keys_rows)}


def few_candidates_hint(head: int, prune, n_users: int = 0, d: int = 0, n_items_local: int = 0, n_splits: int = 0) -> int:
    """The geometry hint (PDA_SWEEP_HUGE | PDA_SWEEP_MANY_CANDIDATES | 0) for pda_score_topk4_* when the caller passes its own n_splits --
    the rules of pda_score_topk_plan (results identical whatever the hint; PDA_SCORE_LISTS=lds|many|huge forces one for A/B measurements
    and tests).  (The name is round 3's: PDA_SWEEP_FEW_CANDIDATES and PDA_SWEEP_WIDE named geometries that round 5 removed.)"""
    forced = os.environ.get("PDA_SCORE_LISTS", "")
    if forced in ("lds", "many", "huge"):
        return {"lds": 0, "many": 8, "huge": 128}[forced]
    if head == HEAD_POP and prune == "order" and d in (64, 128, 256) and n_items_local > 0 and n_splits > 0 \
            and n_splits == huge_splits(n_users, n_items_local, d):
        return 128          # (the caller splits the catalogue as huge_splits says: score_topk_keys with n_splits left to the library)
    if head == HEAD_POP and prune == "order" and n_users >= HUGE_MIN_USERS and d in (64, 128):
        return 128          # PDA_SWEEP_HUGE: 1 024-user workgroups of four 512-register waves, eight MFMAs per LDS read, no test k-step
    early = prune is True or (prune == 1 and prune != "order")
    if d in (64, 128) and not early and (head == HEAD_RAW or not prune):
        # PDA_SWEEP_MANY_CANDIDATES: hundreds of list insertions per user (raw head; popularity head in natural item order) -- 128
        # users per workgroup with eight rescoring waves.  Same box: C3 262 144 users raw 31.3 -> 28.1 ms, natural order 34.2 ->
        # 28.3 ms; C2 raw 3.39 -> 2.10 ms, natural 4.37 -> 2.39 ms.  (Early-terminating raw sweeps: 30.5 vs 37.8 ms, not hinted.)
        return 8
    return 0


def seed_exchange_applies(d: int, K: int, head: int, prune=None, impl: Optional[str] = None) -> bool:
    """Does score_topk_keys(seed_reduce=...) call seed_reduce for these arguments?  A function of arguments that are the same
    on every rank, so that ranks which cannot score (an empty item shard) still know whether to join the two all-reduces."""
    if prune is None:
        prune = prune_default(head, d)
    return (impl or score_impl(d, K, 0)) == "v2" and prune is True and d in (64, 128, 256) and K <= TOPK_K_V4


def seed_rounds(seed_shards: int) -> int:
    """Bisection rounds that tighten the seed of an item-sharded early-terminating sweep (pda_topk_seed_refine; one all-reduce
    of 4 bytes per user each): none up to 3 shards (two shards score 1.07 x the tiles of one GPU with the plain seed), three
    from 4 shards on (config 3: four shards 1.39 x -> 0.98 x, eight shards 1.78 x -> 1.03 x).  PDA_SEED_ROUNDS overrides; the same on every rank."""
    forced = os.environ.get("PDA_SEED_ROUNDS")
    if forced is not None and forced != "":
        return min(4, max(0, int(forced)))
    return 3 if seed_shards >= 4 else 0


def seed_thresholds(seed_shards: int) -> int:
    """Common thresholds of the ONE count exchange that replaces seed_rounds(seed_shards) sequential bisection rounds: the
    2^rounds - 1 interior grid points those rounds walk (7 from four shards on; 0 = the plain seed, no count exchange)."""
    return (1 << seed_rounds(seed_shards)) - 1


def check_order(prep_ord: torch.Tensor, n: int, d: int):
    """Synchronising: raises if the order given to item_prep_ordered was not a permutation of 0..n-1."""
    check(_lib.load().pda_item_prep_ordered_check(ptr(prep_ord), n, d, stream_ptr()), "pda_item_prep_ordered_check")


def hist_reordered(hist: "HistoryCSR", prep_ord: torch.Tensor, order: torch.Tensor, item_offset: int, n: int, d: int) -> torch.Tensor:
    """pda_hist_reorder, cached on the HistoryCSR per visiting order (the order object, not its contents)."""
    cache = hist.__dict__.setdefault("_ord_cache", [])
    for o, off, ind in cache:
        if o is order and off == item_offset:
            return ind
    out = torch.empty_like(hist.indices)
    check(_lib.load().pda_hist_reorder(ptr(prep_ord), n, d, item_offset, ptr(hist.indptr), ptr(hist.indices),
                                       hist.indptr.numel() - 1, ptr(out), stream_ptr()), "pda_hist_reorder")
    del cache[:-3]
    cache.append((order, item_offset, out))
    return out


def mark_modified(*tensors):
    """Tell torch that a kernel wrote these tensors through raw pointers: bumps tensor._version, which is what keys
    the item_prep cache (and autograd's in-place checks)."""
    for t in tensors:
        if t is not None:
            torch.autograd.graph.increment_version(t)


def score_impl(d: int, K: int, item_hi: int) -> str:
    """'v2' = the pre-filtered kernels (bf16 MFMA filter + exact fp32 rescoring: pda_score_topk_v3.hip / _v4.hip; the label
    is historical) where they apply, else 'v1' (exact fp32 MFMA).  Same results.  PDA_SCORE_IMPL=v1|v2 forces one (A/B
    measurements, cross-checks)."""
    import os
    forced = os.environ.get("PDA_SCORE_IMPL", "")
    ok = d in (64, 128, 256) and K <= TOPK_CAP_V2
    if forced == "v1" or not ok:
        return "v1"
    return "v2"


TOPK_CAP_V2 = _lib.TOPK_CAP - 4


_POP_OK = {}           # id(pop) -> (weakref, _version): popularity vectors already checked


def _check_pop(pop_shard: torch.Tensor):
    """The PDA head's filters and bounds are derived for pop >= 0 (it is pop^gamma of a min-max-normalised count,
    MF/train_new_api.py:952-959).  Checked once per tensor version (one device sync).  NaN entries are tolerated: their
    head is NaN, every comparison with it is false, the item is never recommended."""
    hit = _POP_OK.get(id(pop_shard))
    if hit is not None and hit[0]() is pop_shard and hit[1] == pop_shard._version:
        return
    if bool((pop_shard < 0).any()):        # NaN passes: such items simply never rank (the reference's BPRMF-A search feeds NaNs)
        raise ValueError("pop_shard must be >= 0 (it is pop**gamma of a normalised count)")
    for k in [k for k, v in _POP_OK.items() if v[0]() is None]:
        del _POP_OK[k]
    _POP_OK[id(pop_shard)] = (weakref.ref(pop_shard), pop_shard._version)


def prune_default(head: int, d: Optional[int] = None):
    """How the catalogue is swept (results are identical in all three):
        True     visiting order (popular first) + exact early termination -- the default for the popularity-weighted head
        "order"  visiting order, every tile scored (a dense sweep whose thresholds rise early: fewer candidates) -- the
                 default for the raw head at d <= 128 (largest norm first, generation 4: C3 8.3 vs 9.0 ms, C2 3.2 vs 4.0 ms)
        False    natural item order, every tile scored -- the raw head at d = 256 (generation 3: 16.5 vs 22.9 ms)
    PDA_SCORE_PRUNE=0|1|order forces one."""
    import os
    forced = os.environ.get("PDA_SCORE_PRUNE", "")
    if forced in ("0", "1"):
        return forced == "1"
    if forced == "order":
        return "order"
    if head == HEAD_POP:
        return True
    return "order" if d in (64, 128) else False


class SeededCall:
    """One user block of an item-sharded early-terminating sweep between its steps (seeded_begin -> the MAX all-reduce of
    .bounds -> seeded_counts -> the SUM all-reduce of .counts -> seeded_finish): what lets pda_amd.dist run the two collectives
    of block b + 1 on a side stream under the sweep of block b."""
    __slots__ = ("lib", "fnp", "common", "wt", "out", "ws", "bounds", "counts", "n_thr", "K", "nu", "n_splits", "nloc", "stats", "keep")


def seeded_begin(U, I_shard, users, K, head, pop_shard, hist, item_offset=0, n_splits=0, seed_shards=1, prune=True,
                 stats: Optional[dict] = None) -> SeededCall:
    """Step 1: the exact warm-up of this shard (pda_score_topk4_phase_*, phase 1) and the shard's three bounds per user
    (pda_topk_seed_bounds) -> .bounds float32 [3, Bu], to be MAX-reduced over the shards in place."""
    lib = _lib.load()
    bf = I_shard.dtype == torch.bfloat16
    U = _need(U, torch.bfloat16 if bf else torch.float32, "U")
    I_shard = _need(I_shard, torch.bfloat16 if bf else torch.float32, "I_shard")
    users = _need(users, torch.int32, "users")
    pop_shard = _need(pop_shard, torch.float32, "pop_shard", optional=True)
    nu, nloc, d = users.numel(), I_shard.shape[0], I_shard.shape[1]
    if U.shape[1] != d:
        raise ValueError("U and I_shard disagree on embed dim")
    if pop_shard is not None and pop_shard.numel() != nloc:
        raise ValueError("pop_shard must have one entry per local item row")
    if nloc > (1 << 26):
        raise ValueError("seeded item-sharded evaluation: at most 2^26 item rows per shard")
    if head == HEAD_POP and pop_shard is not None:
        _check_pop(pop_shard)
    if hist is not None and hist.indices.numel() == 0:
        hist = None
    c = SeededCall()
    order = visiting_order(I_shard, pop_shard if head == HEAD_POP else None)
    prep = item_prep4(I_shard, pop_shard if head == HEAD_POP else None, order)
    if n_splits <= 0:
        n_splits = lib.pda_score_topk4_auto_splits(nu, nloc, d)
    c.lib, c.K, c.nu, c.n_splits, c.nloc, c.stats = lib, K, nu, n_splits, nloc, stats
    c.out = torch.empty((n_splits, nu, K), dtype=torch.int64, device=U.device)
    c.ws = torch.empty(lib.pda_score_topk4_workspace_bytes(nu, nloc, d, n_splits), dtype=torch.uint8, device=U.device)
    c.fnp = lib.pda_score_topk4_phase_bf16 if bf else lib.pda_score_topk4_phase_f32
    c.common = (ptr(U), ptr(I_shard), ptr(prep), ptr(pop_shard), ptr(users), nu, item_offset, nloc, d,
                ptr(hist.indptr) if hist else None, ptr(hist.indices) if hist else None, hist.mode if hist else 0, K, head,
                1 | few_candidates_hint(head, True, nu, d), n_splits)
    c.keep = (U, I_shard, prep, pop_shard, users, hist)          # the pointers above stay valid until seeded_finish
    # R shards warm up R x 64 warm_tiles items between them: two tiles each on 2 shards, one from 4 shards on
    c.wt = int(os.environ.get("PDA_WARM_TILES", "0")) or max(1, 4 // max(1, seed_shards))
    check(c.fnp(*c.common, 1, c.wt, None, ptr(c.out), ptr(c.ws), stream_ptr()), "pda_score_topk4_phase (warm-up)")
    c.bounds = torch.empty((3, nu), dtype=torch.float32, device=U.device)
    m = -(-K // max(1, seed_shards))
    check(lib.pda_topk_seed_bounds(ptr(c.out), n_splits, nu, K, m, ptr(c.bounds), stream_ptr()), "pda_topk_seed_bounds")
    c.n_thr, c.counts = seed_thresholds(seed_shards), None
    return c


def kth_value(keys: torch.Tensor, pos: int) -> torch.Tensor:
    """pda_topk_kth_value: keys int64 [S, Bu, K] (sorted lists) -> float32 [Bu], the value at rank `pos` (0-based; the largest over
    the S lists, -inf where no list is that long)."""
    keys = _need(keys, torch.int64, "keys")
    S, nu, K = keys.shape
    out = torch.empty(nu, dtype=torch.float32, device=keys.device)
    check(_lib.load().pda_topk_kth_value(ptr(keys), S, nu, K, int(pos), ptr(out), stream_ptr()), "pda_topk_kth_value")
    return out


def remap_key_items(keys: torch.Tensor, gid: torch.Tensor) -> torch.Tensor:
    """pda_topk_remap_items, IN PLACE: the item field of packed keys from local row ids of a gathered table to gid[local]."""
    keys = _need(keys, torch.int64, "keys")
    gid = _need(gid, torch.int32, "gid")
    check(_lib.load().pda_topk_remap_items(ptr(keys), keys.numel(), ptr(gid), gid.numel(), stream_ptr()), "pda_topk_remap_items")
    return keys


def sweep_from_seed(U, I_shard, users, K, head, pop_shard, hist, item_offset, seed: torch.Tensor, n_splits: int = 0, prune="order",
                    stats: Optional[dict] = None) -> torch.Tensor:
    """pda_score_topk4_phase_* with phase 4: the sweep of the whole shard from EMPTY lists against `seed` (float32 [Bu]: a lower
    bound of every user's final K-th value, -inf = none) -- no warm-up on this shard.  -> packed keys int64 [n_splits, Bu, K] of the
    pairs at or above the seed (a list may end shorter than K, or empty)."""
    lib = _lib.load()
    bf = I_shard.dtype == torch.bfloat16
    U = _need(U, torch.bfloat16 if bf else torch.float32, "U")
    I_shard = _need(I_shard, torch.bfloat16 if bf else torch.float32, "I_shard")
    users = _need(users, torch.int32, "users")
    seed = _need(seed, torch.float32, "seed")
    pop_shard = _need(pop_shard, torch.float32, "pop_shard", optional=True)
    nu, nloc, d = users.numel(), I_shard.shape[0], I_shard.shape[1]
    if seed.numel() != nu:
        raise ValueError("seed must have one entry per user of the block")
    if d not in (64, 128, 256) or K > TOPK_K_V4:
        raise ValueError("sweep_from_seed: embed dim 64/128/256 and K <= %d" % TOPK_K_V4)
    if head == HEAD_POP and pop_shard is not None:
        _check_pop(pop_shard)
    if hist is not None and hist.indices.numel() == 0:
        hist = None
    order = visiting_order(I_shard, pop_shard if head == HEAD_POP else None) if prune else None
    prep = item_prep4(I_shard, pop_shard if head == HEAD_POP else None, order)
    if n_splits <= 0:
        n_splits = lib.pda_score_topk4_auto_splits(nu, nloc, d)
        if head == HEAD_POP and prune == "order" and d in (64, 128, 256) and not os.environ.get("PDA_SCORE_LISTS") and pop_shard is not None:
            n_splits = huge_splits(nu, nloc, d) or n_splits
    out = torch.empty((n_splits, nu, K), dtype=torch.int64, device=U.device)
    ws = torch.empty(lib.pda_score_topk4_workspace_bytes(nu, nloc, d, n_splits), dtype=torch.uint8, device=U.device)
    es = (1 if prune is True else 0) | few_candidates_hint(head, prune, nu, d, nloc, n_splits)
    fn = lib.pda_score_topk4_phase_bf16 if bf else lib.pda_score_topk4_phase_f32
    check(fn(ptr(U), ptr(I_shard), ptr(prep), ptr(pop_shard), ptr(users), nu, item_offset, nloc, d,
             ptr(hist.indptr) if hist else None, ptr(hist.indices) if hist else None, hist.mode if hist else 0, K, head, es, n_splits,
             4, 0, ptr(seed), ptr(out), ptr(ws), stream_ptr()), "pda_score_topk4_phase (sweep from a seed)")
    if os.environ.get("PDA_CHECK_SWEEP_ERRORS"):
        err = int(ws[0:4].view(torch.int32)[0])
        if err != 0:
            raise RuntimeError("pda_score_topk4_phase: the sweep reported protocol error %d" % err)
    if stats is not None:
        stats["pairs_rescored"] = ws[4:8].view(torch.int32)
        stats["kernel_id"] = ws[16:20].view(torch.int32)
        stats["error"] = ws[0:4].view(torch.int32)
    return out


def seeded_counts(c: SeededCall) -> Optional[torch.Tensor]:
    """Step 2 (after the MAX all-reduce of c.bounds): this shard's warm-up entries at or above the common thresholds ->
    int32 [n_thr, Bu], to be SUM-reduced in place; None below four shards (plain seed: no second collective)."""
    if c.n_thr <= 0:
        return None
    c.counts = torch.empty((c.n_thr, c.nu), dtype=torch.int32, device=c.bounds.device)
    check(c.lib.pda_topk_seed_counts(ptr(c.out), c.n_splits, c.nu, c.K, ptr(c.bounds), c.n_thr, ptr(c.counts), stream_ptr()),
          "pda_topk_seed_counts")
    return c.counts


def seeded_finish(c: SeededCall) -> torch.Tensor:
    """Step 3 (after the SUM all-reduce of c.counts): the seed per user (pda_topk_seed_pick) and the sweep that prunes against
    it (phase 2).  Returns the shard's packed keys int64 [n_splits, Bu, K]; lists may end shorter than K."""
    seed = torch.empty(c.nu, dtype=torch.float32, device=c.bounds.device)
    check(c.lib.pda_topk_seed_pick(ptr(c.bounds), ptr(c.counts), c.n_thr if c.counts is not None else 0, c.nu, c.K, ptr(seed), stream_ptr()),
          "pda_topk_seed_pick")
    check(c.fnp(*c.common, 2, c.wt, ptr(seed), ptr(c.out), ptr(c.ws), stream_ptr()), "pda_score_topk4_phase (sweep)")
    if c.stats is not None:
        c.stats["tiles_scored"] = c.ws[8:16].view(torch.int64)
        c.stats["pairs_rescored"] = c.ws[4:8].view(torch.int32)
        c.stats["tiles_dense"] = ((c.nloc + 31) // 32) * ((c.nu + 127) // 128)
        c.stats["kernel_id"] = c.ws[16:20].view(torch.int32)
        c.stats["error"] = c.ws[0:4].view(torch.int32)
    out, c.keep = c.out, None
    return out


def score_topk_keys(U, I_shard, users, K=50, head=HEAD_RAW, pop_shard=None, hist: Optional[HistoryCSR] = None,
                    item_offset=0, n_splits=0, out: Optional[torch.Tensor] = None, impl: Optional[str] = None,
                    prune=None, stats: Optional[dict] = None, seed_reduce=None, seed_shards: int = 1, seed_sum=None,
                    warm_tiles: int = 0) -> torch.Tensor:
    """pda_score_topk_f32 / pda_score_topk_prepped_f32 / pda_score_topk_ordered_f32 / pda_score_topk4_* -> packed keys
    int64[n_splits, Bu, K] (uint64 bit patterns), best first.  All of them return the same keys.

    seed_reduce (item-sharded evaluation over `seed_shards` shards, early-terminating sweep): a callable that receives this
    shard's float32 [3, Bu] bounds (pda_topk_seed_bounds: warm-up values at rank K, at rank ceil(K / seed_shards), and minus the
    latter) and turns them IN PLACE into the maximum over all shards (ONE dist.all_reduce MAX).  seed_sum (optional; int32
    [n_thr, Bu] -> the sum over the shards in place, ONE all_reduce SUM): the counts at seed_thresholds(seed_shards) common
    thresholds tighten the bound first.  The sweep prunes against the resulting seed (pda_score_topk4_phase_*); the shard's lists
    may end shorter than K -- merge them with the other shards' lists.  (seeded_begin / seeded_counts / seeded_finish are the
    same three steps as separate calls.)
    warm_tiles (generation 4 only; 0 = the library's default of 4): 64-item tiles per split scored by the exact warm-up kernel --
    an item shard of an R-rank job wants 4 / R (pda_amd.dist): the R warm-ups cover R x 64 x warm_tiles items between them."""
    lib = _lib.load()
    bf = I_shard is not None and I_shard.dtype == torch.bfloat16       # bf16 tables: pda_score_topk_bf16 (both tables bf16)
    U = _need(U, torch.bfloat16 if bf else torch.float32, "U")
    I_shard = _need(I_shard, torch.bfloat16 if bf else torch.float32, "I_shard")
    users = _need(users, torch.int32, "users")
    pop_shard = _need(pop_shard, torch.float32, "pop_shard", optional=True)
    nu, nloc, d = users.numel(), I_shard.shape[0], I_shard.shape[1]
    if bf:
        if d not in (64, 128, 256) or K > TOPK_CAP_V2 or impl == "v1":
            raise ValueError("bf16 tables: embed dim 64/128/256, K <= %d, no v1 entry point" % TOPK_CAP_V2)
        impl = "v2"
    if U.shape[1] != d:
        raise ValueError("U and I_shard disagree on embed dim")
    if pop_shard is not None and pop_shard.numel() != nloc:
        raise ValueError("pop_shard must have one entry per local item row")
    if head == HEAD_POP and pop_shard is not None:       # (a missing pop_shard is the library's PDA_ERR_ARG)
        _check_pop(pop_shard)
    if hist is not None and hist.indices.numel() == 0:
        hist = None                       # an all-empty mask: the kernel must never dereference a 0-length buffer
    n_splits_auto = n_splits <= 0
    out_given = out
    impl = impl or score_impl(d, K, item_offset + nloc)
    if prune is None:
        prune = prune_default(head, d)
    if seed_reduce is not None and seed_exchange_applies(d, K, head, prune, impl):
        if out_given is not None:
            raise ValueError("the seeded sweep allocates its own output")
        c = seeded_begin(U, I_shard, users, K, head, pop_shard, hist, item_offset, n_splits, seed_shards, stats=stats)
        seed_reduce(c.bounds)
        cnt = seeded_counts(c) if seed_sum is not None else None
        if cnt is not None:
            seed_sum(cnt)
        return seeded_finish(c)
    if n_splits_auto and out_given is None and impl == "v2" and funnel_applies(d, K, nu, nloc, head, prune, hist):
        return score_topk_funnel(U, I_shard, users, K, hist, item_offset, stats)
    if n_splits <= 0:
        n_splits = lib.pda_score_topk_auto_splits(nu, nloc)
    if out is None:
        out = torch.empty((n_splits, nu, K), dtype=torch.int64, device=U.device)
    elif out.shape != (n_splits, nu, K) or out.dtype != torch.int64:
        raise ValueError("out must be int64 [n_splits, Bu, K]")
    if impl == "v2" and score_kernel(d, K, nloc, prune, head) == "v4":
        order = visiting_order(I_shard, pop_shard if head == HEAD_POP else None) if prune else None
        prep = item_prep4(I_shard, pop_shard if head == HEAD_POP else None, order)
        forced_env = any(os.environ.get(k) for k in ("PDA_SCORE_LISTS", "PDA_SCORE_KERNEL", "PDA_HUGE_SPLITS"))
        plan = None
        if n_splits_auto and out_given is None and not forced_env and not (head == HEAD_POP and pop_shard is None):
            # the library's own plan (pda_score_topk_plan): item splits and geometry hint
            plan = score_plan(nu, nloc, d, K, head, prune, bf, hist)
            if plan["kernel"] != "v4":
                plan = None
        if plan is not None:
            n_splits = plan["n_splits"]
            out = torch.empty((n_splits, nu, K), dtype=torch.int64, device=U.device)
        elif n_splits_auto and out_given is None:
            n_splits = lib.pda_score_topk4_auto_splits(nu, nloc, d)
            if head == HEAD_POP and prune == "order" and d in (64, 128, 256) and not os.environ.get("PDA_SCORE_LISTS") and pop_shard is not None:
                n_splits = huge_splits(nu, nloc, d) or n_splits          # the huge geometry on a block that does not fill the chip by itself
            out = torch.empty((n_splits, nu, K), dtype=torch.int64, device=U.device)
        ws = torch.empty(lib.pda_score_topk4_workspace_bytes(nu, nloc, d, n_splits), dtype=torch.uint8, device=U.device)
        hint = plan["early_stop"] if plan is not None else ((1 if prune is True else 0) | few_candidates_hint(head, prune, nu, d, nloc, n_splits))
        es = hint | ((min(4, max(0, int(warm_tiles))) & 7) << 4)
        if os.environ.get("PDA_WARM_PER_SPLIT"):      # A/B measurements and cross-checks: every item split warms up on its own tiles (before round 4)
            es |= SWEEP_WARM_PER_SPLIT
        fn = lib.pda_score_topk4_bf16 if bf else lib.pda_score_topk4_f32
        check(fn(ptr(U), ptr(I_shard), ptr(prep), ptr(pop_shard), ptr(users), nu, item_offset, nloc, d,
                 ptr(hist.indptr) if hist else None, ptr(hist.indices) if hist else None, hist.mode if hist else 0,
                 K, head, es, n_splits, ptr(out), ptr(ws), stream_ptr()), "pda_score_topk4")
        if os.environ.get("PDA_CHECK_SWEEP_ERRORS"):          # (tests: synchronising) a bounded wait of the sweep ran out, or the huge geometry's self-check failed
            err = int(ws[0:4].view(torch.int32)[0])
            if err != 0:
                raise RuntimeError("pda_score_topk4: the sweep reported protocol error %d" % err)
        if stats is not None:
            stats["tiles_scored"] = ws[8:16].view(torch.int64)
            stats["pairs_rescored"] = ws[4:8].view(torch.int32)
            stats["tiles_dense"] = ((nloc + 31) // 32) * ((nu + 127) // 128)
            stats["kernel_id"] = ws[16:20].view(torch.int32)     # written by the sweep kernel itself: see kernel_identity()
            stats["huge_entries"] = ws[20:24].view(torch.int32)  # huge geometry: entries of its asm loop, summed over the waves (1 per wave + 1 per flagged half-tile)
            stats["error"] = ws[0:4].view(torch.int32)           # 0, or which bounded wait of the sweep ran out (1 .. 4: hand-over words; 5, 6: huge geometry)
        return out
    if impl == "v2" and prune:
        prep, order = item_prep_ordered(I_shard, pop_shard if head == HEAD_POP else None)
        hist_ord = hist_reordered(hist, prep, order, item_offset, nloc, d) if hist else None
        ws = torch.empty(lib.pda_score_topk_workspace_bytes(nu), dtype=torch.uint8, device=U.device)
        fn = lib.pda_score_topk_ordered_bf16 if bf else lib.pda_score_topk_ordered_f32
        check(fn(ptr(U), ptr(I_shard), ptr(prep), ptr(pop_shard), ptr(users), nu, item_offset,
                                             nloc, d, ptr(hist.indptr) if hist else None,
                                             ptr(hist.indices) if hist else None, ptr(hist_ord), hist.mode if hist else 0,
                                             K, head, 0 if prune == "order" else 1, n_splits, ptr(out), ptr(ws), stream_ptr()),
              "pda_score_topk_ordered")
        if stats is not None:            # device scalar (no sync here): item tiles scored, summed over workgroups
            stats["tiles_scored"] = ws[8:16].view(torch.int64)
            stats["pairs_rescored"] = ws[4:8].view(torch.int32)        # v3 kernel only (0 otherwise)
            stats["tiles_dense"] = ((nloc + 31) // 32) * ((nu + 127) // 128)
            stats["kernel_id"] = ws[16:20].view(torch.int32)
        return out
    if impl == "v2":
        prep = item_prep(I_shard)
        ws = torch.empty(lib.pda_score_topk_workspace_bytes(nu), dtype=torch.uint8, device=U.device)   # per call: re-entrant
        fn = lib.pda_score_topk_bf16 if bf else lib.pda_score_topk_prepped_f32
        check(fn(ptr(U), ptr(I_shard), ptr(prep), ptr(pop_shard), ptr(users), nu, item_offset,
                                             nloc, d, ptr(hist.indptr) if hist else None,
                                             ptr(hist.indices) if hist else None, hist.mode if hist else 0, K, head,
                                             n_splits, ptr(out), ptr(ws), stream_ptr()), "pda_score_topk_prepped_f32")
        return out
    check(lib.pda_score_topk_f32(ptr(U), ptr(I_shard), ptr(pop_shard), ptr(users), nu, item_offset, nloc, d,
                                 ptr(hist.indptr) if hist else None, ptr(hist.indices) if hist else None,
                                 hist.mode if hist else 0, K, head, n_splits, ptr(out), stream_ptr()),
          "pda_score_topk_f32")
    return out


# ---- the funnel (pda_score_topk7_*): the raw head on large user blocks ------------------------------------------------------------
# Measured (config 3, same box): 262 144 users 22.3 vs 29.7 ms for generation 4's many-candidates geometry, 65 536 users 6.4 vs 7.3 ms; config 2 (50 000 users
# x 20 000 items, d = 64) 3.3 vs 2.1 ms -- a funnel is ~25 launches whose per-row work does not shrink with the catalogue.
FUNNEL_MIN_USERS = 1             # (round 6; 1 024 before: a partly filled 1 024-user tile still wins -- 256 users x 200 000 items 0.27 - 0.32 vs 2.0 - 2.2 ms for generation 4)
FUNNEL_MIN_ITEMS = 4096           # (round 6: the funnel's own minimum; 8 192 items x 4 096 users 0.32 vs 0.43 ms for generation 4)
FUNNEL_BLOCK_ROW_MAX_USERS = 16384        # a mask by BLOCK ROW (the reference's per-block COO triple): the exact fallback sweeps the whole block again if a row fails
FUNNEL_SMALL_ITEMS, FUNNEL_SMALL_MAX_USERS = 20000, 16384     # catalogues below 20 000 items: up to 16 384 users (16 384 items x 65 536 users: 2.1 vs 1.9 ms for generation 4)
_FUNNEL_ORDER = {}               # (n, device) -> a fixed pseudo-random permutation (the object is what the prep cache keys on)


def funnel_order(I_shard: torch.Tensor) -> torch.Tensor:
    """The visiting order of a funnel: a RANDOM permutation of the shard (seeded by its size: reproducible).  The ranks of the funnel's
    thresholds assume that the items seen so far are a uniform sample of the catalogue; results are exact for any order."""
    key = (I_shard.shape[0], str(I_shard.device))
    o = _FUNNEL_ORDER.get(key)
    if o is None:
        g = torch.Generator(device="cpu").manual_seed(0x5EED + I_shard.shape[0])
        o = torch.randperm(I_shard.shape[0], generator=g).to(torch.int32).to(I_shard.device)
        _FUNNEL_ORDER[key] = o
    return o


FUNNEL_WORKSPACE_BUDGET = 24 << 30      # PDA_FUNNEL_WORKSPACE_BUDGET (include/pda_hip.h)


def funnel_applies(d: int, K: int, nu: int, nloc: int, head: int, prune, hist: Optional[HistoryCSR]) -> bool:
    """Does score_topk_keys serve this call with the funnel?  (PDA_SCORE_FUNNEL=0 | 1 forces it off / on wherever it can run.)"""
    can = head == HEAD_RAW and d in (64, 128, 256) and K <= TOPK_K_V4 and 4096 <= nloc <= (1 << 26) and prune is not True \
        and (hist is None or hist.mode == HIST_BY_USER_ID or nu <= FUNNEL_BLOCK_ROW_MAX_USERS)
    forced = os.environ.get("PDA_SCORE_FUNNEL", "")
    if forced == "0" or not can:
        return False
    if forced == "1":
        return True
    return nu >= FUNNEL_MIN_USERS and nloc >= FUNNEL_MIN_ITEMS and (nloc >= FUNNEL_SMALL_ITEMS or nu <= FUNNEL_SMALL_MAX_USERS) \
        and not os.environ.get("PDA_SCORE_KERNEL") and not os.environ.get("PDA_SCORE_LISTS") \
        and _lib.load().pda_score_topk7_workspace_bytes(nu, nloc, d) <= FUNNEL_WORKSPACE_BUDGET     # (~27 KB per user: pda_score_topk_plan's rule)


def score_topk_funnel(U, I_shard, users, K=50, hist: Optional[HistoryCSR] = None, item_offset=0, stats: Optional[dict] = None) -> torch.Tensor:
    """pda_score_topk7_f32 / _bf16: the raw head through the funnel -> packed keys int64 [1, Bu, K] (the item splits are merged inside)."""
    lib = _lib.load()
    bf = I_shard.dtype == torch.bfloat16
    U = _need(U, torch.bfloat16 if bf else torch.float32, "U")
    I_shard = _need(I_shard, torch.bfloat16 if bf else torch.float32, "I_shard")
    users = _need(users, torch.int32, "users")
    nu, nloc, d = users.numel(), I_shard.shape[0], I_shard.shape[1]
    if hist is not None and hist.indices.numel() == 0:
        hist = None
    prep = item_prep7(I_shard, funnel_order(I_shard))
    out = torch.empty((1, nu, K), dtype=torch.int64, device=U.device)
    nbytes = lib.pda_score_topk7_workspace_bytes(nu, nloc, d)
    if nbytes == 0:
        raise ValueError("the funnel: embed dim 64 / 128 / 256")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=U.device)
    fn = lib.pda_score_topk7_bf16 if bf else lib.pda_score_topk7_f32
    check(fn(ptr(U), ptr(I_shard), ptr(prep), ptr(users), nu, item_offset, nloc, d, ptr(hist.indptr) if hist else None,
             ptr(hist.indices) if hist else None, hist.mode if hist else 0, K, HEAD_RAW, ptr(out), ptr(ws), stream_ptr()), "pda_score_topk7")
    if stats is not None:
        stats["pairs_rescored"] = ws[4:8].view(torch.int32)
        stats["kernel_id"] = ws[16:20].view(torch.int32)
        stats["error"] = ws[0:4].view(torch.int32)
        stats["fallback_rows"] = ws[24:28].view(torch.int32)      # rows served by the exact fallback (generation 4) inside the call
        stats["workspace"] = ws                                   # (tools/check_funnel.py reads the rows' state out of it)
    return out


GEOMETRY_NAMES = {0: "lds", 3: "many", 4: "huge", 7: "funnel"}          # (1, 2, 5, 6: the geometries round 5 removed)


def kernel_identity(word) -> dict:
    """Decodes the word the sweep kernels write at workspace + 16 (stats["kernel_id"]): which kernel generation, geometry and
    template instance a score_topk_keys call actually ran.  Synchronising when given the device tensor.  0 = no pre-filtered
    sweep ran (every split ended inside its warm-up, or generation 1)."""
    w = int(word) & 0xFFFFFFFF
    if w == 0:
        return {"generation": 0}
    return {"generation": w >> 28, "geometry": GEOMETRY_NAMES.get((w >> 8) & 15, "?") if (w >> 28) == 4 else None,
            ("early_stop" if (w >> 28) == 4 else "visiting_order"): bool((w >> 12) & 1), "head": (w >> 13) & 1, "bf16": bool((w >> 14) & 1), "d": (w & 15) * 64}


def topk_merge(keys: torch.Tensor, users=None, hist: Optional[HistoryCSR] = None, want="idx_val"):
    """pda_topk_merge.  keys int64[R, Bu, K].  want: 'keys' -> int64[Bu,K];  'idx_val' -> (int32, float32)."""
    lib = _lib.load()
    keys = _need(keys, torch.int64, "keys")
    R, nu, K = keys.shape
    dev = keys.device
    out_keys = out_idx = out_val = None
    if want == "keys":
        out_keys = torch.empty((nu, K), dtype=torch.int64, device=dev)
    else:
        out_idx = torch.empty((nu, K), dtype=torch.int32, device=dev)
        out_val = torch.empty((nu, K), dtype=torch.float32, device=dev)
    if users is not None:
        users = _need(users, torch.int32, "users")
    if hist is not None and hist.indices.numel() == 0:
        hist = None
    check(lib.pda_topk_merge(ptr(keys), R, nu, K, ptr(out_keys), ptr(out_idx), ptr(out_val), ptr(users),
                             ptr(hist.indptr) if hist else None, ptr(hist.indices) if hist else None,
                             hist.mode if hist else 0, stream_ptr()), "pda_topk_merge")
    return out_keys if want == "keys" else (out_idx, out_val)


def recommend_topk(U, I_shard, users, K=50, head=HEAD_RAW, pop_shard=None, hist=None, item_offset=0,
                   n_splits=0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Single-GPU convenience: score + mask + top-K + split merge -> (idx int32[Bu,K], val f32[Bu,K])."""
    keys = score_topk_keys(U, I_shard, users, K, head, pop_shard, hist, item_offset, n_splits)
    return topk_merge(keys, users, hist, want="idx_val")


def bpr_step(U, I, users, pos, neg, pos_pop=None, neg_pop=None, *, regs: float, reg_div: float, lr: float = 0.0,
             mode: int = UPD_NONE, grads_out=None, gU=None, gI=None, loss_acc: Optional[torch.Tensor] = None,
             grouped: bool = False, users_distinct: bool = False):
    """pda_bpr_step_f32.  grads_out = (g_user, g_pos, g_neg) float32 [B,d] or None.  grouped: the batch went through
    group_triplets_by_pos / sort_triplets_by_pos (equal positives adjacent); otherwise PDA_UPD_ANY_ORDER is set.
    users_distinct: the caller's assertion that no user id repeats in the batch (the sampler contract, pda_amd.sampler
    .distinct_users): PDA_UPD_USERS_DISTINCT, the fused step's user rows take plain stores instead of atomics."""
    lib = _lib.load()
    U = _need(U, torch.float32, "U")
    I = _need(I, torch.float32, "I")
    users, pos, neg = (_need(t, torch.int32, n) for t, n in ((users, "users"), (pos, "pos"), (neg, "neg")))
    pos_pop = _need(pos_pop, torch.float32, "pos_pop", optional=True)
    neg_pop = _need(neg_pop, torch.float32, "neg_pop", optional=True)
    B, d = users.numel(), U.shape[1]
    if pos.numel() != B or neg.numel() != B:
        raise ValueError("users/pos/neg must have the same length")
    gu = gp = gn = None
    if grads_out is not None:
        gu, gp, gn = (_need(t, torch.float32, "grads_out") for t in grads_out)
    gU = _need(gU, torch.float32, "gU", optional=True)
    gI = _need(gI, torch.float32, "gI", optional=True)
    loss_acc = _need(loss_acc, torch.float32, "loss_acc", optional=True)
    check(lib.pda_bpr_step_f32(ptr(U), ptr(I), ptr(users), ptr(pos), ptr(neg), ptr(pos_pop), ptr(neg_pop), B, d,
                               float(regs), float(reg_div), float(lr),
                               mode | (0 if grouped else UPD_ANY_ORDER) | (UPD_USERS_DISTINCT if users_distinct and mode == UPD_SGD_FUSED else 0),
                               ptr(gu), ptr(gp), ptr(gn), ptr(gU), ptr(gI), ptr(loss_acc), stream_ptr()), "pda_bpr_step_f32")
    if mode == UPD_SGD_FUSED:
        mark_modified(U, I)


def sgd_step_exact(U, I, users, pos, neg, pos_pop=None, neg_pop=None, *, regs: float, reg_div: float, lr: float,
                   loss_acc: Optional[torch.Tensor] = None, scratch=None):
    """The exact mini-batch SGD step: pda_bpr_step_f32(PDA_UPD_NONE) writes the per-occurrence gradients of the whole batch
    against the unchanged tables, pda_sgd_apply_f32 scatters them.  `scratch` = (g_user, g_pos, g_neg) float32 [B, d] to
    reuse between steps (allocated when None).  Returns the scratch tuple."""
    lib = _lib.load()
    B, d = users.numel(), U.shape[1]
    if scratch is None:
        scratch = tuple(torch.empty((B, d), dtype=torch.float32, device=U.device) for _ in range(3))
    bpr_step(U, I, users, pos, neg, pos_pop, neg_pop, regs=regs, reg_div=reg_div, mode=UPD_NONE, grads_out=scratch, loss_acc=loss_acc)
    check(lib.pda_sgd_apply_f32(ptr(U), ptr(I), ptr(users), ptr(pos), ptr(neg), ptr(scratch[0]), ptr(scratch[1]), ptr(scratch[2]),
                                B, d, float(lr), stream_ptr()), "pda_sgd_apply_f32")
    mark_modified(U, I)
    return scratch


PLAN_LDS_MAX_B = 4096     # pda_triplet_plan sorts a batch inside one workgroup's LDS up to here
_PLAN_WS = {}


def triplet_plan_bytes(B: int) -> int:
    return _lib.load().pda_triplet_plan_bytes(B)


def triplet_plan(users, pos, neg, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pda_triplet_plan: users / pos / neg int32 [B] or [n, B] -> uint8 [n, pda_triplet_plan_bytes(B)] (one plan per batch; the
    plan of batch j is out[j]).  B <= 4096: one workgroup per batch; larger batches: pda_triplet_plan_large, a device-wide sort
    per batch (same plan bytes)."""
    lib = _lib.load()
    users, pos, neg = (_need(t, torch.int32, n) for t, n in ((users, "users"), (pos, "pos"), (neg, "neg")))
    B = users.shape[-1]
    n = users.numel() // B
    nb = lib.pda_triplet_plan_bytes(B)
    if out is None:
        out = torch.empty((n, nb), dtype=torch.uint8, device=users.device)
    elif out.shape != (n, nb) or out.dtype != torch.uint8 or not out.is_contiguous():
        raise ValueError("out must be contiguous uint8 [n_batches, pda_triplet_plan_bytes(B)]")
    if B > PLAN_LDS_MAX_B:
        key = (B, users.device)
        ws = _PLAN_WS.get(key)
        if ws is None:
            ws = _PLAN_WS[key] = torch.empty(lib.pda_triplet_plan_large_workspace_bytes(B), dtype=torch.uint8, device=users.device)
        u2, p2, n2 = users.reshape(n, B), pos.reshape(n, B), neg.reshape(n, B)
        for j in range(n):
            check(lib.pda_triplet_plan_large(ptr(u2[j]), ptr(p2[j]), ptr(n2[j]), B, ptr(out[j]), ptr(ws), stream_ptr()), "pda_triplet_plan_large")
        return out
    check(lib.pda_triplet_plan(ptr(users), ptr(pos), ptr(neg), B, n, ptr(out), stream_ptr()), "pda_triplet_plan")
    return out


def plan_header(plan: torch.Tensor):
    """Host view of a plan's header (synchronising; tests and diagnostics): (segments, a user occurs twice, 2B, B)."""
    return tuple(int(x) for x in plan.reshape(-1)[:16].view(torch.int32).cpu())


def bpr_step_plan(U, I, users, pos, neg, pos_pop=None, neg_pop=None, *, regs: float, reg_div: float, lr: float, plan: torch.Tensor,
                  scratch: Optional[torch.Tensor] = None, exact: bool = True, loss_acc: Optional[torch.Tensor] = None,
                  U_master=None, I_master=None) -> Optional[torch.Tensor]:
    """pda_bpr_step_plan_f32 / _bf16 (U, I bf16 with fp32 U_master / I_master): the exact mini-batch SGD step of a batch with
    DISTINCT users, two launches, no atomics (exact=False: one launch, plain stores on user rows and once-referenced item rows,
    atomics on shared item rows; fp32 tables only).  plan = triplet_plan(users, pos, neg)[j].  Returns the scratch buffer for reuse."""
    lib = _lib.load()
    bf = U.dtype == torch.bfloat16
    users, pos, neg = (_need(t, torch.int32, n) for t, n in ((users, "users"), (pos, "pos"), (neg, "neg")))
    pos_pop = _need(pos_pop, torch.float32, "pos_pop", optional=True)
    neg_pop = _need(neg_pop, torch.float32, "neg_pop", optional=True)
    plan = _need(plan, torch.uint8, "plan")
    B, d = users.numel(), U.shape[1]
    if plan.numel() != lib.pda_triplet_plan_bytes(B):
        raise ValueError("plan does not belong to a batch of %d triplets" % B)
    if exact:
        ns = lib.pda_bpr_step_plan_scratch_bytes(B, d) // 4
        if scratch is None or scratch.numel() != ns:
            scratch = torch.empty(ns, dtype=torch.float32, device=U.device)
    loss_acc = _need(loss_acc, torch.float32, "loss_acc", optional=True)
    if bf:
        U, I = _need(U, torch.bfloat16, "U"), _need(I, torch.bfloat16, "I")
        U_master, I_master = _need(U_master, torch.float32, "U_master"), _need(I_master, torch.float32, "I_master")
        if not exact:
            raise ValueError("bf16 tables take the exact planned step only")
        check(lib.pda_bpr_step_plan_bf16(ptr(U), ptr(I), ptr(U_master), ptr(I_master), ptr(users), ptr(pos), ptr(neg), ptr(pos_pop), ptr(neg_pop),
                                         B, d, float(regs), float(reg_div), float(lr), ptr(plan), ptr(scratch), ptr(loss_acc), stream_ptr()),
              "pda_bpr_step_plan_bf16")
        mark_modified(U, I, U_master, I_master)
    else:
        U, I = _need(U, torch.float32, "U"), _need(I, torch.float32, "I")
        check(lib.pda_bpr_step_plan_f32(ptr(U), ptr(I), ptr(users), ptr(pos), ptr(neg), ptr(pos_pop), ptr(neg_pop), B, d, float(regs),
                                        float(reg_div), float(lr), ptr(plan), ptr(scratch), 1 if exact else 0, ptr(loss_acc), stream_ptr()),
              "pda_bpr_step_plan_f32")
        mark_modified(U, I)
    return scratch


def bpr_step_bf16(U16, I16, users, pos, neg, pos_pop=None, neg_pop=None, *, regs: float, reg_div: float, lr: float = 0.0,
                  mode: int = UPD_NONE, U_master=None, I_master=None, grads_out=None, gU=None, gI=None,
                  loss_acc: Optional[torch.Tensor] = None, refresh: bool = True):
    """pda_bpr_step_bf16 (+ pda_refresh_rows_bf16 of the touched rows after a fused SGD update)."""
    lib = _lib.load()
    U16 = _need(U16, torch.bfloat16, "U16")
    I16 = _need(I16, torch.bfloat16, "I16")
    users, pos, neg = (_need(t, torch.int32, n) for t, n in ((users, "users"), (pos, "pos"), (neg, "neg")))
    pos_pop = _need(pos_pop, torch.float32, "pos_pop", optional=True)
    neg_pop = _need(neg_pop, torch.float32, "neg_pop", optional=True)
    U_master = _need(U_master, torch.float32, "U_master", optional=True)
    I_master = _need(I_master, torch.float32, "I_master", optional=True)
    B, d = users.numel(), U16.shape[1]
    gu = gp = gn = None
    if grads_out is not None:
        gu, gp, gn = (_need(t, torch.float32, "grads_out") for t in grads_out)
    gU = _need(gU, torch.float32, "gU", optional=True)
    gI = _need(gI, torch.float32, "gI", optional=True)
    loss_acc = _need(loss_acc, torch.float32, "loss_acc", optional=True)
    check(lib.pda_bpr_step_bf16(ptr(U16), ptr(I16), ptr(U_master), ptr(I_master), ptr(users), ptr(pos), ptr(neg), ptr(pos_pop),
                                ptr(neg_pop), B, d, float(regs), float(reg_div), float(lr), mode | UPD_ANY_ORDER, ptr(gu), ptr(gp), ptr(gn),
                                ptr(gU), ptr(gI), ptr(loss_acc), stream_ptr()), "pda_bpr_step_bf16")
    if mode == UPD_SGD_FUSED:
        mark_modified(U_master, I_master)
        if refresh:
            refresh_rows_bf16(U_master, U16, users)
            refresh_rows_bf16(I_master, I16, pos)
            refresh_rows_bf16(I_master, I16, neg)


def refresh_rows_bf16(master, shadow, rows=None):
    """pda_refresh_rows_bf16: shadow[rows] = bf16(master[rows]) (RNE); rows None = the whole table."""
    lib = _lib.load()
    master = _need(master, torch.float32, "master")
    shadow = _need(shadow, torch.bfloat16, "shadow")
    rows = _need(rows, torch.int32, "rows", optional=True)
    n = rows.numel() if rows is not None else master.shape[0]
    check(lib.pda_refresh_rows_bf16(ptr(master), ptr(shadow), ptr(rows), n, master.shape[1], stream_ptr()), "pda_refresh_rows_bf16")
    mark_modified(shadow)


def bpr_step_shard(U, I_shard, item_offset: int, users, pos, neg, pos_pop=None, neg_pop=None, *, regs: float, reg_div: float,
                   mean_div: float, lr: float, g_user: torch.Tensor, loss_acc: Optional[torch.Tensor] = None,
                   gI_shard: Optional[torch.Tensor] = None):
    """pda_bpr_step_shard_f32: one rank's part of an item-parallel SGD step.  pos/neg are GLOBAL ids inside
    [item_offset, item_offset + I_shard.shape[0]); g_user float32 [B_local, >=d] (row stride = g_user.stride(0))."""
    lib = _lib.load()
    U = _need(U, torch.float32, "U")
    I_shard = _need(I_shard, torch.float32, "I_shard")
    users, pos, neg = (_need(t, torch.int32, n) for t, n in ((users, "users"), (pos, "pos"), (neg, "neg")))
    pos_pop = _need(pos_pop, torch.float32, "pos_pop", optional=True)
    neg_pop = _need(neg_pop, torch.float32, "neg_pop", optional=True)
    if g_user.dtype != torch.float32 or not g_user.is_cuda:
        raise ValueError("g_user must be a float32 cuda tensor")
    if loss_acc is not None and (loss_acc.dtype != torch.float32 or not loss_acc.is_cuda or loss_acc.numel() != 3 or loss_acc.stride(0) != 1):
        raise ValueError("loss_acc must be 3 consecutive float32 on the device")
    B, d = users.numel(), U.shape[1]
    if g_user.shape != (B, d) or g_user.stride(1) != 1:
        raise ValueError("g_user must be float32 [B_local, d] with unit inner stride")
    check(lib.pda_bpr_step_shard_f32(ptr(U), ptr(I_shard), int(item_offset), ptr(users), ptr(pos), ptr(neg), ptr(pos_pop),
                                     ptr(neg_pop), B, d, float(regs), float(reg_div), float(mean_div), float(lr),
                                     ptr(g_user), g_user.stride(0), ptr(_need(gI_shard, torch.float32, "gI_shard", optional=True)),
                                     ptr(loss_acc), stream_ptr()), "pda_bpr_step_shard_f32")
    if gI_shard is None:
        mark_modified(I_shard)


def apply_user_grads(U, users, g, lr: float):
    """pda_apply_user_grads_f32: U[users[i]] -= lr * g[i].  g float32 [n, d] whose rows may be strided (a column slice
    of the packed exchange buffer); its storage must start 16-byte aligned."""
    lib = _lib.load()
    U = _need(U, torch.float32, "U")
    users = _need(users, torch.int32, "users")
    if g.dtype != torch.float32 or not g.is_cuda or g.stride(1) != 1:
        raise ValueError("g must be a float32 cuda tensor with unit inner stride")
    n, d = g.shape
    check(lib.pda_apply_user_grads_f32(ptr(U), ptr(users), ptr(g), n, d, g.stride(0), float(lr), stream_ptr()),
          "pda_apply_user_grads_f32")
    mark_modified(U)


def score_dense(U, I, users, head: int, pop=None, items=None) -> torch.Tensor:
    """pda_score_dense_f32: the ratings matrix float32 [len(users), n_items] (batch_ratings / condition_ratings of the reference), every entry
    the exact fmaf chain of the top-K kernels.  items: int32 row ids of I or None (all rows); pop: float32 per returned column."""
    U, I = _need(U, torch.float32, "U"), _need(I, torch.float32, "I")
    users = _need(users, torch.int32, "users")
    items = _need(items, torch.int32, "items", optional=True)
    pop = _need(pop, torch.float32, "pop", optional=True)
    n = I.shape[0] if items is None else items.numel()
    if pop is not None and pop.numel() != n:
        raise ValueError("pop holds one value per returned column")
    out = torch.empty((users.numel(), n), dtype=torch.float32, device=U.device)
    check(_lib.load().pda_score_dense_f32(ptr(U), ptr(I), ptr(pop), ptr(users), users.numel(), ptr(items), n, U.shape[1], int(head), ptr(out), stream_ptr()),
          "pda_score_dense_f32")
    return out


def adam_lr_t(lr: float, t: int, beta1=ADAM_BETA1, beta2=ADAM_BETA2) -> float:
    """lr * sqrt(1-beta2^t) / (1-beta1^t)  [TF-ext AdamOptimizer._apply_sparse_shared]."""
    return lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)


def adam_dense_sweep2(var_a, m_a, v_a, g_a, var_b, m_b, v_b, g_b, lr_t: float, beta1=ADAM_BETA1, beta2=ADAM_BETA2, eps=ADAM_EPS):
    """pda_adam_dense_sweep2_f32: the dense-decay Adam sweep over both tables of the model in one launch."""
    lib = _lib.load()
    for t in (var_a, m_a, v_a, g_a, var_b, m_b, v_b, g_b):
        _need(t, torch.float32, "adam state")
    check(lib.pda_adam_dense_sweep2_f32(ptr(var_a), ptr(m_a), ptr(v_a), ptr(g_a), var_a.numel(), ptr(var_b), ptr(m_b), ptr(v_b), ptr(g_b),
                                        var_b.numel(), lr_t, beta1, beta2, eps, stream_ptr()), "pda_adam_dense_sweep2_f32")
    mark_modified(var_a)
    mark_modified(var_b)


def adam_touched_bitmaps(n_users: int, n_items: int, device):
    """Zeroed "touched by this step" bitmaps for adam_mark_rows / adam_dense_sweep3 (one bit per table row)."""
    return (torch.zeros((n_users + 31) // 32, dtype=torch.int32, device=device), torch.zeros((n_items + 31) // 32, dtype=torch.int32, device=device))


def adam_mark_rows(users, pos, neg, touched_u, touched_i):
    """pda_adam_mark_rows: the batch's rows into the bitmaps."""
    if pos.numel() != users.numel() or neg.numel() != users.numel():
        raise ValueError("users/pos/neg must have the same length")
    check(_lib.load().pda_adam_mark_rows(ptr(_need(users, torch.int32, "users")), ptr(_need(pos, torch.int32, "pos")), ptr(_need(neg, torch.int32, "neg")),
                                         users.numel(), ptr(touched_u), ptr(touched_i), stream_ptr()), "pda_adam_mark_rows")


def adam_dense_sweep3(var_a, m_a, v_a, g_a, touched_a, var_b, m_b, v_b, g_b, touched_b, lr_t: float, beta1=ADAM_BETA1, beta2=ADAM_BETA2, eps=ADAM_EPS):
    """pda_adam_dense_sweep3_f32: the dense-decay Adam sweep over both tables without reading the gradient tables outside the marked rows
    (bit-identical to adam_dense_sweep2); clears the marks."""
    lib = _lib.load()
    for t in (var_a, m_a, v_a, g_a, var_b, m_b, v_b, g_b):
        _need(t, torch.float32, "adam state")
    check(lib.pda_adam_dense_sweep3_f32(ptr(var_a), ptr(m_a), ptr(v_a), ptr(g_a), var_a.shape[0], ptr(touched_a), ptr(var_b), ptr(m_b), ptr(v_b), ptr(g_b),
                                        var_b.shape[0], ptr(touched_b), var_a.shape[1], lr_t, beta1, beta2, eps, stream_ptr()), "pda_adam_dense_sweep3_f32")
    mark_modified(var_a)
    mark_modified(var_b)


ADAM_CACHE_AUTO, ADAM_CACHE_RESIDENT, ADAM_CACHE_STREAM = 0, 1, 2


def adam_row_tags(n_users: int, n_items: int, device):
    """Zeroed per-row step tags for adam_step / adam_dense_sweep4 (i32 per table row; a row is 'touched by step t' iff its tag == t)."""
    return torch.zeros(n_users, dtype=torch.int32, device=device), torch.zeros(n_items, dtype=torch.int32, device=device)


def adam_step(U, mU, vU, gU, tagU, I, mI, vI, gI, tagI, users, pos, neg, pos_pop=None, neg_pop=None, *, regs: float, reg_div: float, step: int,
              lr_t: float, beta1=ADAM_BETA1, beta2=ADAM_BETA2, eps=ADAM_EPS, grouped: bool = False, users_distinct: bool = False,
              cache_policy: int = ADAM_CACHE_AUTO, loss_acc: Optional[torch.Tensor] = None):
    """pda_adam_step_f32: one reference train step (gradients of the batch + TF-1.14 dense-decay Adam over both tables) in two launches.
    step >= 1 is the step number (the tag the touched rows get); gU / gI must be zero off the rows of the running step (they are, when only
    this function writes them)."""
    lib = _lib.load()
    for t in (U, mU, vU, gU, I, mI, vI, gI):
        _need(t, torch.float32, "adam state")
    users, pos, neg = (_need(t, torch.int32, n) for t, n in ((users, "users"), (pos, "pos"), (neg, "neg")))
    tagU, tagI = _need(tagU, torch.int32, "tagU"), _need(tagI, torch.int32, "tagI")
    pos_pop = _need(pos_pop, torch.float32, "pos_pop", optional=True)
    neg_pop = _need(neg_pop, torch.float32, "neg_pop", optional=True)
    B, d = users.numel(), U.shape[1]
    if pos.numel() != B or neg.numel() != B:
        raise ValueError("users/pos/neg must have the same length")
    if tagU.numel() != U.shape[0] or tagI.numel() != I.shape[0]:
        raise ValueError("tagU / tagI hold one int32 per table row")
    loss_acc = _need(loss_acc, torch.float32, "loss_acc", optional=True)
    check(lib.pda_adam_step_f32(ptr(U), ptr(mU), ptr(vU), ptr(gU), ptr(tagU), U.shape[0], ptr(I), ptr(mI), ptr(vI), ptr(gI), ptr(tagI), I.shape[0],
                                ptr(users), ptr(pos), ptr(neg), ptr(pos_pop), ptr(neg_pop), B, d, float(regs), float(reg_div), int(step), float(lr_t),
                                beta1, beta2, eps, (0 if grouped else UPD_ANY_ORDER) | (UPD_USERS_DISTINCT if users_distinct else 0),
                                int(cache_policy), ptr(loss_acc), stream_ptr()), "pda_adam_step_f32")
    mark_modified(U, I)


def adam_dense_sweep4(var_a, m_a, v_a, g_a, tag_a, var_b, m_b, v_b, g_b, tag_b, tag: int, lr_t: float, beta1=ADAM_BETA1, beta2=ADAM_BETA2, eps=ADAM_EPS,
                      cache_policy: int = ADAM_CACHE_AUTO):
    """pda_adam_dense_sweep4_f32: the sweep of adam_step alone (the gradients and the tags came from elsewhere)."""
    lib = _lib.load()
    for t in (var_a, m_a, v_a, g_a, var_b, m_b, v_b, g_b):
        _need(t, torch.float32, "adam state")
    check(lib.pda_adam_dense_sweep4_f32(ptr(var_a), ptr(m_a), ptr(v_a), ptr(g_a), var_a.shape[0], ptr(_need(tag_a, torch.int32, "tag_a")), ptr(var_b), ptr(m_b),
                                        ptr(v_b), ptr(g_b), var_b.shape[0], ptr(_need(tag_b, torch.int32, "tag_b")), var_a.shape[1], int(tag), float(lr_t),
                                        beta1, beta2, eps, int(cache_policy), stream_ptr()), "pda_adam_dense_sweep4_f32")
    mark_modified(var_a)
    mark_modified(var_b)


def adam_dense_sweep(var, m, v, g, lr_t: float, beta1=ADAM_BETA1, beta2=ADAM_BETA2, eps=ADAM_EPS):
    lib = _lib.load()
    for t, n in ((var, "var"), (m, "m"), (v, "v"), (g, "g")):
        _need(t, torch.float32, n)
    check(lib.pda_adam_dense_sweep_f32(ptr(var), ptr(m), ptr(v), ptr(g), var.numel(), lr_t, beta1, beta2, eps,
                                       stream_ptr()), "pda_adam_dense_sweep_f32")
    mark_modified(var)


def adam_rows(var, m, v, g, rows, lr_t: float, beta1=ADAM_BETA1, beta2=ADAM_BETA2, eps=ADAM_EPS):
    lib = _lib.load()
    rows = _need(rows, torch.int32, "rows")
    check(lib.pda_adam_rows_f32(ptr(var), ptr(m), ptr(v), ptr(g), ptr(rows), rows.numel(), var.shape[1], lr_t, beta1,
                                beta2, eps, stream_ptr()), "pda_adam_rows_f32")
    mark_modified(var)


class LazyAdamState:
    """The bookkeeping of the exact lazy dense-decay Adam (pda_adam_lazy_f32): per-row `last` steps and the device table of
    bias-corrected rates lr_tab[k] = float32(adam_lr_t(lr, k)) -- the value the dense sweep receives as its lr_t argument."""

    def __init__(self, n_users: int, n_items: int, lr: float, device, beta1=ADAM_BETA1, beta2=ADAM_BETA2, eps=ADAM_EPS, fast: bool = False):
        self.fast = bool(fast)              # PDA_ADAM_REPLAY_FAST: catch-up to 1e-6 instead of bit for bit (see include/pda_hip.h)
        self.lastU = torch.zeros(n_users, dtype=torch.int32, device=device)
        self.lastI = torch.zeros(n_items, dtype=torch.int32, device=device)
        self.lr, self.beta1, self.beta2, self.eps = lr, beta1, beta2, eps
        self.lr_tab = torch.zeros(0, dtype=torch.float32, device=device)
        self.synced = 0                     # the step every row is known to be current for

    def rates(self, t: int) -> torch.Tensor:
        """lr_tab with at least t + 1 entries (grown geometrically; computed in float64 like adam_lr_t, rounded once)."""
        if self.lr_tab.numel() < t + 1:
            # the very doubles the dense path passes as lr_t (adam_lr_t), rounded to float32 once -- as the ctypes call does
            n = max(1024, 2 * (t + 1))
            tab = [0.0] + [adam_lr_t(self.lr, k, self.beta1, self.beta2) for k in range(1, n)]
            self.lr_tab = torch.tensor(tab, dtype=torch.float64).to(torch.float32).to(self.lastU.device)
        return self.lr_tab


def adam_lazy(phase: int, st: LazyAdamState, U, mU, vU, gU, I, mI, vI, gI, users, pos, neg, t: int):
    """pda_adam_lazy_f32: phase 0 = the batch rows up to step t - 1 (before the forward pass of step t), phase 1 = step t on
    the batch rows with their summed gradients (after pda_bpr_step_f32 in PDA_UPD_DENSE_GRAD mode)."""
    lib = _lib.load()
    for x in (U, mU, vU, gU, I, mI, vI, gI):
        _need(x, torch.float32, "adam state")
    for x in (users, pos, neg):
        _need(x, torch.int32, "batch rows")
    tab = st.rates(t)
    check(lib.pda_adam_lazy_f32(phase | (_lib.ADAM_REPLAY_FAST if st.fast else 0), ptr(U), ptr(mU), ptr(vU), ptr(gU), ptr(st.lastU), ptr(I), ptr(mI), ptr(vI), ptr(gI), ptr(st.lastI),
                                ptr(users), ptr(pos), ptr(neg), users.numel(), U.shape[1], t, ptr(tab), st.beta1, st.beta2, st.eps,
                                stream_ptr()), "pda_adam_lazy_f32")
    mark_modified(U)
    mark_modified(I)


def adam_lazy_dev(phase: int, st: LazyAdamState, U, mU, vU, gU, I, mI, vI, gI, users, pos, neg, t_dev: torch.Tensor, parity: int, n_tab: int):
    """pda_adam_lazy_dev_f32: adam_lazy with the step in device memory -- t_dev int32 [2]: slot `parity` is read, phase 1 stores
    t + 1 into slot 1 - parity (the caller alternates the parity from step to step: an even number of steps per captured graph).
    n_tab: steps the rate table must cover (built once, before the capture)."""
    lib = _lib.load()
    for x in (U, mU, vU, gU, I, mI, vI, gI):
        _need(x, torch.float32, "adam state")
    for x in (users, pos, neg):
        _need(x, torch.int32, "batch rows")
    t_dev = _need(t_dev, torch.int32, "t_dev")
    if t_dev.numel() != 2:
        raise ValueError("t_dev must be int32 [2]")
    tab = st.rates(n_tab)
    check(lib.pda_adam_lazy_dev_f32(phase | (_lib.ADAM_REPLAY_FAST if st.fast else 0), ptr(U), ptr(mU), ptr(vU), ptr(gU), ptr(st.lastU), ptr(I), ptr(mI),
                                    ptr(vI), ptr(gI), ptr(st.lastI), ptr(users), ptr(pos), ptr(neg), users.numel(), U.shape[1],
                                    ptr(t_dev[parity:parity + 1]), ptr(t_dev[1 - parity:2 - parity]), ptr(tab), tab.numel(), st.beta1, st.beta2, st.eps,
                                    stream_ptr()), "pda_adam_lazy_dev_f32")
    mark_modified(U)
    mark_modified(I)


def adam_lazy_sync(st: LazyAdamState, U, mU, vU, I, mI, vI, t: int):
    """pda_adam_lazy_sync_f32 on both tables: every row current for step t (a no-op when nothing is behind)."""
    if st.synced >= t:
        return
    lib = _lib.load()
    tab = st.rates(t)
    fn = lib.pda_adam_lazy_sync_fast_f32 if st.fast else lib.pda_adam_lazy_sync_f32
    for var, m, v, last in ((U, mU, vU, st.lastU), (I, mI, vI, st.lastI)):
        check(fn(ptr(var), ptr(m), ptr(v), ptr(last), var.shape[0], var.shape[1], t, ptr(tab), st.beta1, st.beta2,
                                         st.eps, stream_ptr()), "pda_adam_lazy_sync_f32")
        mark_modified(var)
    st.synced = t


def metrics_sums(topk, tgt_indptr, tgt_indices, Ks, sums: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pda_metrics: float64 [4, len(Ks)] sums of precision, recall, ndcg, hit over the rows (added to `sums`)."""
    lib = _lib.load()
    topk = _need(topk, torch.int32, "topk")
    tgt_indptr = _need(tgt_indptr, torch.int64, "tgt_indptr")
    tgt_indices = _need(tgt_indices, torch.int32, "tgt_indices")
    Ks = _need(Ks, torch.int32, "Ks")
    if sums is None:
        sums = torch.zeros((4, Ks.numel()), dtype=torch.float64, device=topk.device)
    check(lib.pda_metrics(ptr(topk), topk.shape[0], topk.shape[1], ptr(tgt_indptr), ptr(tgt_indices), ptr(Ks),
                          Ks.numel(), ptr(sums), stream_ptr()), "pda_metrics")
    return sums


def sample_triplets(train_indptr, train_indices, B: int, *, seed: int, step: int, users=None, user_pool=None,
                    n_pool: int = 0, train_slots=None, neg_range=(0, 0), pop_matrix=None, sort_by_pos: bool = False):
    """pda_sample_triplets -> (users, pos, neg, pos_pop|None, neg_pop|None), all on device."""
    lib = _lib.load()
    dev = train_indptr.device
    gen = users is None
    if gen:
        users = torch.empty(B, dtype=torch.int32, device=dev)
    pos = torch.empty(B, dtype=torch.int32, device=dev)
    neg = torch.empty(B, dtype=torch.int32, device=dev)
    pp = pn = None
    n_slots = 0
    if pop_matrix is not None:
        pop_matrix = _need(pop_matrix, torch.float32, "pop_matrix")
        n_slots = pop_matrix.shape[1]
        pp = torch.empty(B, dtype=torch.float32, device=dev)
        pn = torch.empty(B, dtype=torch.float32, device=dev)
    check(lib.pda_sample_triplets(ptr(users), int(gen), ptr(user_pool), int(n_pool), B, ptr(train_indptr),
                                  ptr(train_indices), ptr(train_slots), int(neg_range[0]), int(neg_range[1]),
                                  ptr(pop_matrix), n_slots, seed & (2 ** 64 - 1), step, ptr(pos), ptr(neg), ptr(pp),
                                  ptr(pn), stream_ptr()), "pda_sample_triplets")
    if sort_by_pos:
        group_triplets_by_pos(users, pos, neg, pp, pn)
    return users, pos, neg, pp, pn


def sample_triplets_into(out, train_indptr, train_indices, *, seed: int, step_dev: torch.Tensor, user_pool=None, n_pool: int = 0,
                          train_slots=None, neg_range=(0, 0), pop_matrix=None, sort_by_pos: bool = False, advance: bool = True,
                          parity: Optional[int] = None):
    """Graph-capturable sampler: writes one batch into the preallocated `out` = (users, pos, neg, pos_pop|None, neg_pop|None),
    taking the step from device memory.  step_dev int64[1]: read, and with `advance` incremented by a pda_counter_add launch.
    step_dev int64[2] + parity p: the step is read from slot p and step + 1 stored to slot 1 - p by the sampler itself (no
    extra launch); the caller alternates p from call to call (an even number of calls per captured graph)."""
    lib = _lib.load()
    users, pos, neg, pp, pn = out
    n_slots = pop_matrix.shape[1] if pop_matrix is not None else 0
    if parity is None:
        src, nxt = step_dev, None
    else:
        if step_dev.numel() != 2:
            raise ValueError("parity mode needs a two-slot step counter")
        src, nxt = step_dev[parity:parity + 1], step_dev[1 - parity:2 - parity]
    check(lib.pda_sample_triplets_dev(ptr(users), 1, ptr(user_pool), int(n_pool), users.numel(), ptr(train_indptr),
                                      ptr(train_indices), ptr(train_slots), int(neg_range[0]), int(neg_range[1]),
                                      ptr(pop_matrix), n_slots, seed & (2 ** 64 - 1), ptr(src), ptr(nxt) if advance else None,
                                      ptr(pos), ptr(neg), ptr(pp), ptr(pn), stream_ptr()), "pda_sample_triplets_dev")
    if sort_by_pos:
        group_triplets_by_pos(users, pos, neg, pp, pn)
    if advance and parity is None:
        check(lib.pda_counter_add(ptr(step_dev), 1, stream_ptr()), "pda_counter_add")
    return out


def sample_batches_into(out, train_indptr, train_indices, *, seed: int, step_dev: torch.Tensor, parity: int, user_pool=None, n_pool: int = 0,
                        train_slots=None, neg_range=(0, 0), pop_matrix=None, group_by_pos: bool = False):
    """pda_sample_batches_dev: `out` = (users, pos, neg, pos_pop|None, neg_pop|None) as [n, B] tensors; row j receives the batch
    of step step_dev[parity] + j (bit for bit what sample_triplets_into draws for that step); step_dev[1 - parity] receives
    step + n.  Graph-capturable (alternate `parity` from call to call, an even number of calls per captured graph)."""
    lib = _lib.load()
    users, pos, neg, pp, pn = out
    if users.dim() != 2 or step_dev.numel() != 2:
        raise ValueError("sample_batches_into wants [n, B] buffers and the two-slot step counter")
    n, B = users.shape
    n_slots = pop_matrix.shape[1] if pop_matrix is not None else 0
    check(lib.pda_sample_batches_dev(ptr(users), 1, ptr(user_pool), int(n_pool), B, n, ptr(train_indptr), ptr(train_indices), ptr(train_slots),
                                     int(neg_range[0]), int(neg_range[1]), ptr(pop_matrix), n_slots, seed & (2 ** 64 - 1),
                                     ptr(step_dev[parity:parity + 1]), ptr(step_dev[1 - parity:2 - parity]), ptr(pos), ptr(neg), ptr(pp), ptr(pn),
                                     1 if group_by_pos else 0, stream_ptr()), "pda_sample_batches_dev")
    return out


def bpr_step_and_sample(U, I, users, pos, neg, pos_pop, neg_pop, *, regs: float, reg_div: float, lr: float, next_out,
                        train_indptr, train_indices, seed: int, step_dev: torch.Tensor, parity: int, user_pool=None,
                        n_pool: int = 0, train_slots=None, neg_range=(0, 0), pop_matrix=None, mode: int = UPD_SGD_FUSED,
                        loss_acc: Optional[torch.Tensor] = None, grouped: bool = False):
    """pda_bpr_step_sample_f32: the fused SGD step on (users, pos, neg, ...) and, in spare workgroups of the same launch, the
    sampler of the NEXT batch into `next_out` = (users, pos, neg, pos_pop|None, neg_pop|None) -- a second set of batch
    buffers.  step_dev int64[2] + parity as in sample_triplets_into (slot `parity` is read, 1 - parity receives step + 1).
    Graph-capturable; equivalent to bpr_step followed by sample_triplets_into."""
    lib = _lib.load()
    if step_dev.numel() != 2:
        raise ValueError("bpr_step_and_sample needs the two-slot step counter")
    nu, npos, nneg, npp, npn = next_out
    n_slots = pop_matrix.shape[1] if pop_matrix is not None else 0
    job = _lib.SampleJob(ptr(nu), 1, ptr(user_pool), int(n_pool), nu.numel(), ptr(train_indptr), ptr(train_indices), ptr(train_slots),
                         int(neg_range[0]), int(neg_range[1]), ptr(pop_matrix), n_slots, seed & (2 ** 64 - 1),
                         ptr(step_dev[parity:parity + 1]), ptr(step_dev[1 - parity:2 - parity]), ptr(npos), ptr(nneg), ptr(npp), ptr(npn))
    if loss_acc is None:
        loss_acc = torch.zeros(3, dtype=torch.float32, device=U.device)
    m = int(mode) | (0 if grouped else UPD_ANY_ORDER)
    check(lib.pda_bpr_step_sample_f32(ptr(U), ptr(I), ptr(users), ptr(pos), ptr(neg), ptr(pos_pop), ptr(neg_pop), users.numel(),
                                      U.shape[1], float(regs), float(reg_div), float(lr), m, ptr(loss_acc), C.byref(job), stream_ptr()),
          "pda_bpr_step_sample_f32")
    return loss_acc


def bpr_train_steps(U, I, bufs, n_steps: int, *, regs: float, reg_div: float, lr: float, train_indptr, train_indices, seed: int,
                    step_ctr: torch.Tensor, user_pool=None, n_pool: int = 0, train_slots=None, neg_range=(0, 0), pop_matrix=None,
                    loss_acc: Optional[torch.Tensor] = None, loss_steps: Optional[torch.Tensor] = None, grouped: bool = False,
                    barrier_ws: Optional[torch.Tensor] = None):
    """pda_bpr_train_steps_f32: n_steps fused SGD steps in one launch.  bufs = two sets (users, pos, neg, pos_pop|None,
    neg_pop|None) of batch buffers; set 0 holds the first batch, set n_steps & 1 the next one afterwards.  step_ctr int64[1]
    (device): the sampler step of the first batch drawn inside, advanced by n_steps.  Returns (loss tensor, barrier_ws) --
    barrier_ws[1] != 0 (checked by the caller when it next synchronises) means the loop was abandoned."""
    lib = _lib.load()
    n_slots = pop_matrix.shape[1] if pop_matrix is not None else 0
    jobs = []
    for (nu, npos, nneg, npp, npn) in bufs:
        jobs.append(_lib.SampleJob(ptr(nu), 1, ptr(user_pool), int(n_pool), nu.numel(), ptr(train_indptr), ptr(train_indices), ptr(train_slots),
                                   int(neg_range[0]), int(neg_range[1]), ptr(pop_matrix), n_slots, seed & (2 ** 64 - 1), None, None,
                                   ptr(npos), ptr(nneg), ptr(npp), ptr(npn)))
    if loss_acc is None and loss_steps is None:
        loss_acc = torch.zeros(3, dtype=torch.float32, device=U.device)
    if barrier_ws is None:
        barrier_ws = torch.zeros(2, dtype=torch.int32, device=U.device)
    if step_ctr.dtype != torch.int64 or step_ctr.numel() != 1:
        raise ValueError("step_ctr must be int64[1] on the device")
    m = int(UPD_SGD_FUSED) | (0 if grouped else UPD_ANY_ORDER)
    check(lib.pda_bpr_train_steps_f32(ptr(U), ptr(I), U.shape[1], float(regs), float(reg_div), float(lr), m, C.byref(jobs[0]), C.byref(jobs[1]),
                                      ptr(step_ctr), int(n_steps), ptr(loss_acc), ptr(loss_steps), ptr(barrier_ws), stream_ptr()),
          "pda_bpr_train_steps_f32")
    mark_modified(U)
    mark_modified(I)
    return (loss_steps if loss_steps is not None else loss_acc), barrier_ws


def group_triplets_by_pos(users, pos, neg, pos_pop=None, neg_pop=None):
    """pda_group_triplets_by_pos (in place): equal positives become contiguous.  Batches above 4096 triplets are left alone."""
    lib = _lib.load()
    if users.numel() > 4096:
        return
    check(lib.pda_group_triplets_by_pos(ptr(users), ptr(pos), ptr(neg), ptr(pos_pop), ptr(neg_pop), users.numel(), stream_ptr()),
          "pda_group_triplets_by_pos")


def sort_triplets_by_pos(users, pos, neg, pos_pop=None, neg_pop=None):
    """pda_sort_triplets_by_pos (in place).  Batches above 4096 triplets are left as they are."""
    lib = _lib.load()
    if users.numel() > 4096:
        return
    check(lib.pda_sort_triplets_by_pos(ptr(users), ptr(pos), ptr(neg), ptr(pos_pop), ptr(neg_pop), users.numel(), stream_ptr()),
          "pda_sort_triplets_by_pos")


def measured_peaks(device=None, mfma_iters: int = 4000, copy_mb: int = 1024) -> dict:
    """pda_peak_mfma_bf16 / pda_peak_copy timed with HIP events (best of 3): the roofs of THIS box, beside the datasheet's."""
    lib = _lib.load()
    dev = device or torch.device("cuda")
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    n = copy_mb * (1 << 20) // 4
    src, dst = torch.ones(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev)
    best_mfma = best_copy = float("inf")
    for _ in range(4):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        check(lib.pda_peak_mfma_bf16(ptr(sink), mfma_iters, stream_ptr()), "pda_peak_mfma_bf16")
        e[1].record()
        check(lib.pda_peak_copy(ptr(src), ptr(dst), n, stream_ptr()), "pda_peak_copy")
        e[2].record()
        torch.cuda.synchronize()
        best_mfma, best_copy = min(best_mfma, e[0].elapsed_time(e[1])), min(best_copy, e[1].elapsed_time(e[2]))
    # the same register-operand loop with every operand 1.0, and the sweep's own kind of loop (B from the LDS, tiles by LDS-DMA)
    # on random and on constant bf16 rows: the second data points behind `peak_measured`
    best_const = best_lds = best_lds_const = float("inf")
    n_blk = 3000
    nb = (n_blk + 1024) * 19456
    rows = torch.randint(-(1 << 15), (1 << 15) - 1, (nb // 2,), dtype=torch.int16, device=dev)
    rows = ((rows & 0x7F) | ((0x3B + (torch.arange(nb // 2, device=dev, dtype=torch.int16) & 3)) << 7) | (rows & -0x8000)).contiguous()   # |x| ~ 0.1 .. 1
    rows_c = torch.full_like(rows, 0x3C00)
    scratch = torch.zeros(4, dtype=torch.int32, device=dev)
    for _ in range(3):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        check(lib.pda_peak_mfma_bf16_const(ptr(sink), mfma_iters, stream_ptr()), "pda_peak_mfma_bf16_const")
        e[1].record()
        check(lib.pda_peak_mfma_lds_bf16(ptr(rows), nb, ptr(scratch), n_blk, stream_ptr()), "pda_peak_mfma_lds_bf16")
        e[2].record()
        check(lib.pda_peak_mfma_lds_bf16(ptr(rows_c), nb, ptr(scratch), n_blk, stream_ptr()), "pda_peak_mfma_lds_bf16")
        e[3].record()
        torch.cuda.synchronize()
        best_const = min(best_const, e[0].elapsed_time(e[1]))
        best_lds, best_lds_const = min(best_lds, e[1].elapsed_time(e[2])), min(best_lds_const, e[2].elapsed_time(e[3]))
    lds_fl = lib.pda_peak_mfma_lds_flops_per_launch(n_blk)
    return {"bf16_mfma_TFLOPs": lib.pda_peak_mfma_flops_per_launch(mfma_iters) / (best_mfma * 1e-3) / 1e12,
            "bf16_mfma_constant_operands_TFLOPs": lib.pda_peak_mfma_flops_per_launch(mfma_iters) / (best_const * 1e-3) / 1e12,
            "bf16_mfma_lds_fed_TFLOPs": lds_fl / (best_lds * 1e-3) / 1e12,
            "bf16_mfma_lds_fed_constant_rows_TFLOPs": lds_fl / (best_lds_const * 1e-3) / 1e12,
            "hbm_copy_GBs": 2.0 * 4.0 * n / (best_copy * 1e-3) / 1e9,
            "note": "bf16 MFMA: 2 waves per SIMD x 4 accumulator chains, register operands, random mantissas (power-limited clock) / every operand "
                    "1.0; lds_fed: the sweep's own block statement -- B from the LDS, one ds_read_b128 per MFMA, four LDS-DMA loader waves streaming "
                    "tiles, no hand-over (executed flops: 18 MFMAs per block, of which 16 algorithmic) on random / constant bf16 rows; "
                    "HBM: float4 copy of %d MiB, read + write bytes" % copy_mb}


def unpack_keys(keys: torch.Tensor):
    """Host-side helper for tests: packed int64 keys -> (idx int64, val float32); empty (0) -> (-1, -inf)."""
    k = keys.cpu().numpy().view("uint64")
    import numpy as np
    hi = (k >> np.uint64(32)).astype(np.uint32)
    lo = (k & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    bits = np.where(hi & np.uint32(0x80000000), hi ^ np.uint32(0x80000000), ~hi).astype(np.uint32)
    val = bits.view(np.float32).copy()
    idx = (np.uint32(0xFFFFFFFF) - lo).astype(np.int64)
    empty = k == 0
    idx[empty] = -1
    val[empty] = -np.inf
    return idx, val
