"""Douban-shaped synthetic workloads: the processed Douban data is absent from the reference
tree (data/douban/douban.zip is a missing blob), so every benchmark and scale test uses this generator.

Everything is produced directly in HBM with torch (plumbing, not product): N(0, 0.1^2) embeddings (Xavier
at 1M users gives |w| < 0.0025 and degenerate rankings), Zipf(1.0) item popularity over a random item
permutation, clipped log-normal history lengths, ten time slots of which the first nine are train slots,
per-slot popularity by the pop_pre.py:31-42 recipe raised to gamma.  Seeds follow the reference's habit
(MF/train_new_api.py:934-936): 2020 for data, 2021 for weights.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Optional

import torch

CONFIGS = {
    # name: (n_users, n_items, embed, mean_hist)      BASELINE.json configs[0..4]
    # c1: the shape of the processed Douban data the reference trains on (README.md:41,69; ~6.7 M train pairs) -- the data
    #     itself is a missing blob in the reference tree
    "c1": (47_890, 26_047, 64, 140),
    "c2": (50_000, 20_000, 64, 150),
    "c3": (1_000_000, 200_000, 128, 50),
    # c5shard: ONE rank's share of config 5 (10 M users x 2 M items, d = 256, bf16 tables, item-sharded over 8 GPUs):
    #     250 000 item rows against the WHOLE replicated user table (10 M rows: 5.1 GB of bf16) and its 600 M-entry history CSR (2.4 GB of item ids: byte offsets beyond 2^31; round 6; rounds 2 - 5
    #     kept a 1 M-row replica, "c5shard1m": byte offsets of user rows and CSR entries never passed 2^31 there)
    "c5shard": (10_000_000, 250_000, 256, 60),
    "c5shard1m": (1_000_000, 250_000, 256, 50),
    "tiny": (4_000, 3_000, 64, 30),
}


@dataclasses.dataclass
class Workload:
    name: str
    n_users: int
    n_items: int
    d: int
    gamma: float
    U: torch.Tensor                 # f32 [n_users, d]
    I: torch.Tensor                 # f32 [n_items, d]
    hist_indptr: torch.Tensor       # i64 [n_users+1]   train CSR by user id, items sorted ascending
    hist_indices: torch.Tensor      # i32 [nnz]
    hist_slots: torch.Tensor        # i32 [nnz]         time slot (0..T-2) of each train interaction
    pop_train: torch.Tensor         # f32 [n_items, T-1]  pop^gamma per train slot  (train_new_api.py:988-990)
    pop_last: torch.Tensor          # f32 [n_items]       last-stage pop^gamma       (:954-955)
    n_train: int


def _gen(seed: int, device) -> torch.Generator:
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return g


def make_workload(name: str = "c2", device="cuda", gamma: float = 0.22, n_slots: int = 10,
                  n_users: Optional[int] = None, n_items: Optional[int] = None, d: Optional[int] = None,
                  mean_hist: Optional[int] = None, table_dtype: torch.dtype = torch.float32) -> Workload:
    """table_dtype=torch.bfloat16: U and I are returned as bf16 tables (config 5)."""
    cu, ci, cd, ch = CONFIGS[name]
    n_users, n_items, d, mean_hist = n_users or cu, n_items or ci, d or cd, mean_hist or ch
    dev = torch.device(device)
    gd, gw = _gen(2020, dev), _gen(2021, dev)

    U = torch.randn(n_users, d, generator=gw, device=dev) * 0.1
    I = torch.randn(n_items, d, generator=gw, device=dev) * 0.1
    if table_dtype != torch.float32:          # (at once: 10 M x 256 fp32 rows are 10 GB the caller never sees)
        U, I = U.to(table_dtype), I.to(table_dtype)

    # history lengths: clipped log-normal with the requested mean (sigma 0.8)
    sigma = 0.8
    mu = math.log(mean_hist) - 0.5 * sigma * sigma
    lens = torch.exp(torch.randn(n_users, generator=gd, device=dev) * sigma + mu).clamp_(1, min(4 * mean_hist + 200, n_items // 2)).long()
    indptr = torch.zeros(n_users + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=indptr[1:])
    nnz = int(indptr[-1])

    # item draw: Zipf(1.0) over a random permutation, by inverse CDF
    w = 1.0 / torch.arange(1, n_items + 1, device=dev, dtype=torch.float64)
    cdf = torch.cumsum(w / w.sum(), 0).float()
    perm = torch.randperm(n_items, generator=gd, device=dev)
    draws = torch.searchsorted(cdf, torch.rand(nnz, generator=gd, device=dev)).clamp_(max=n_items - 1)
    items = perm[draws]
    rows = torch.repeat_interleave(torch.arange(n_users, device=dev), lens)
    slots = torch.randint(0, n_slots - 1, (nnz,), generator=gd, device=dev, dtype=torch.int32)
    order = torch.argsort(rows * n_items + items)              # sort items inside each row (kernel contract)
    items, slots = items[order].int().contiguous(), slots[order].contiguous()
    del rows, order, draws

    # per-slot popularity, pop_pre.py:31-42: (cnt+1)/(total+n_item), min-max per slot, then ^gamma
    cnt = torch.bincount(slots.long() * n_items + items.long(), minlength=n_slots * n_items).double().view(n_slots, n_items)
    cnt[n_slots - 1] = cnt[n_slots - 2] * 0.9 + cnt[n_slots - 3] * 0.1   # a test-stage slot, shaped like its neighbours
    tot = cnt.sum(1, keepdim=True)
    pop = (cnt + 1.0) / (tot + n_items)
    pop = (pop - pop.min(1, keepdim=True).values) / (pop.max(1, keepdim=True).values - pop.min(1, keepdim=True).values)
    pop_all = pop.t().contiguous()                              # [I, T] like item_pop_seq_ori2.txt
    pop_train = pop_all[:, :-1].pow(gamma).float().contiguous()
    pop_last = pop_all[:, -2].pow(gamma).float().contiguous()
    if table_dtype != torch.float32:
        U, I = U.to(table_dtype), I.to(table_dtype)
    return Workload(name, n_users, n_items, d, gamma, U, I, indptr, items, slots, pop_train, pop_last, nnz)


def write_dataset(root: str, n_users=600, n_items=400, n_slots=10, mean_hist=25, seed=2020):
    """Write a small Douban-shaped dataset in the reference's on-disk formats (host numpy; for tests and
    for trying the CLI):  train.txt, train_with_time.txt, valid.txt, test.txt, t_0..t_{T-1}.txt  and, through the
    pop_pre restatement, item_pop_seq_ori2.txt.  Slots 0..T-2 are train, slot T-1 is split 70/30 by user into
    test/valid (data/douban/douban_split.ipynb, cells 16 and 26).  Returns the directory."""
    import os

    import numpy as np

    from . import pop_pre
    os.makedirs(root, exist_ok=True)
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, n_items + 1)
    perm = rng.permutation(n_items)
    p = np.empty(n_items)
    p[perm] = w / w.sum()
    rows = []   # (u, i, t)
    for u in range(n_users):
        n = int(np.clip(rng.lognormal(np.log(mean_hist) - 0.32, 0.8), 3, n_items // 2))
        items = rng.choice(n_items, size=n, replace=False, p=p)
        slots = rng.integers(0, n_slots, n)
        slots[:2] = [0, n_slots - 2]                         # every user has train rows in >= 2 slots
        rows += [(u, int(i), int(t)) for i, t in zip(items, slots)]
    rows = np.array(rows)
    rows[:n_items, 1] = np.arange(n_items)                   # every item id occurs (pop_pre counts distinct ids)
    train = rows[rows[:, 2] < n_slots - 1]
    last = rows[rows[:, 2] == n_slots - 1]
    is_test = rng.random(n_users) < 0.7

    def write_lists(name, triples):
        by = {}
        for u, i, _ in triples:
            by.setdefault(int(u), []).append(int(i))
        with open(os.path.join(root, name), "w") as f:
            for u, items in by.items():
                f.write(" ".join(str(x) for x in [u] + items) + "\n")

    write_lists("train.txt", train)
    write_lists("test.txt", last[is_test[last[:, 0]]])
    write_lists("valid.txt", last[~is_test[last[:, 0]]])
    with open(os.path.join(root, "train_with_time.txt"), "w") as f:
        for u, i, t in train:
            f.write("%d %d %d 5\n" % (u, i, t))
    for t in range(n_slots):
        by = {}
        for u, i, _ in rows[rows[:, 2] == t]:
            by.setdefault(int(i), []).append(int(u))
        with open(os.path.join(root, "t_%d.txt" % t), "w") as f:
            for i, us in by.items():
                f.write(" ".join(str(x) for x in [i] + us) + "\n")
    pop = pop_pre.compute_popularity(pop_pre.read_stage_counts(root, n_slots), n_item=n_items)
    pop_pre.write_popularity(root, pop)
    return root
