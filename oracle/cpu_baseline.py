"""CPU baseline = torch-CPU fp32 restatement of the *exact reference op sequence* (BASELINE.md section 3).

TEST/BENCH INFRASTRUCTURE ONLY (see oracle/pda_oracle.py header): used by bench.py's `cpu_baseline` leg and by
tests; never imported by pda_amd.  The literal TF-1.14 CPU path cannot run here or on the GPU box, so every
number produced by this file is labelled kind="port".

eval  (MF/model_api.py:62,113; MF/train_new_api.py:594-612, 780-794), per 2048-user block:
      R = U[u] @ I.T  (materialised [Bu, I])  ->  elu  ->  +1  ->  * pop  ->  scatter -inf at history  ->  topk(50)
train (MF/model_api.py:51-53,102-121,83): index_select x3 -> dots -> (ELU+1)*pop -> -mean(log(sigmoid+1e-10))
      + regs*sum(l2)/B -> autograd -> Adam with dense decay over both full tables [TF-ext].
"""
from __future__ import annotations

import time

import torch
import torch.nn.functional as F


def eval_block(U, I, pop, users, coo_rows, coo_cols, K=50, rec_type="condition"):
    R = U.index_select(0, users) @ I.t()                         # tf.matmul, model_api.py:62
    if rec_type != "main_branch":
        R = (F.elu(R) + 1.0) * pop.unsqueeze(0)                  # model_api.py:113 / train_new_api.py:601-602
    R[coo_rows, coo_cols] = float("-inf")                        # tf.sparse.add(-inf), :597,603,608
    return torch.topk(R, K, dim=1, sorted=True).indices          # tf.nn.top_k, :598,604,609


def eval_block_native_topk(U, I, pop, users, coo_rows, coo_cols, K=50, rec_type="condition"):
    """The same block with the selection done by the native CPU top-K (oracle_arg_topk_2d: a heap per row, rows over all
    OpenMP threads -- BASELINE.md section 3 "CPU-native-topk", the algorithm class of util/cython/include/arg_topk.h:15-45)."""
    from . import c_oracle
    R = U.index_select(0, users) @ I.t()
    if rec_type != "main_branch":
        R = (F.elu(R) + 1.0) * pop.unsqueeze(0)
    R[coo_rows, coo_cols] = float("-inf")
    return c_oracle.arg_topk_2d(R.numpy(), K)


_POOLS = {}


def eval_block_slabbed(U, I, pop, users, coo_rows, coo_cols, K=50, rec_type="condition", threads=None, slab=64):
    """The same op sequence -- matmul, elu + 1, * pop, -inf at the train items, topk -- with the 2048-user reference block cut
    into `slab`-row pieces handed to `threads` host threads, each running single-threaded torch ops on its piece (torch releases
    the GIL inside an op).  eval_block above leaves the threading to torch's intra-op pool, which parallelises the matmul but
    runs the scatter and most of topk on one core: 128 cores give 2 x one core there.  Same results row for row (the ops are
    row-independent).  coo_rows must be ascending (they are: the block's rows are built user by user).
    The caller sets torch.set_num_threads(1) around the timed region (time_eval does when block_fn is this function)."""
    from concurrent.futures import ThreadPoolExecutor
    threads = threads or (torch.get_num_interop_threads() and __import__("os").cpu_count()) or 1
    pool = _POOLS.get(threads)
    if pool is None:
        pool = _POOLS[threads] = ThreadPoolExecutor(max_workers=threads)
    n = users.numel()
    bounds = torch.searchsorted(coo_rows, torch.arange(0, n + slab, slab))
    out = torch.empty((n, K), dtype=torch.int64)

    def piece(s):
        lo, hi = s * slab, min(n, (s + 1) * slab)
        a, b = int(bounds[s]), int(bounds[s + 1])
        out[lo:hi] = eval_block(U, I, pop, users[lo:hi], coo_rows[a:b] - lo, coo_cols[a:b], K, rec_type)
    list(pool.map(piece, range((n + slab - 1) // slab)))
    return out


def eval_block_blocked(U, I, pop, users, coo_rows, coo_cols, K=50, rec_type="condition", threads=None, slab=64, chunk=16384):
    """eval_block_slabbed with the catalogue cut as well: every host thread takes 64 user rows and walks the items in chunks of
    16 384 -- matmul, elu + 1, * pop, -inf at the train items, topk(K) of the 4 MB piece (measured best of 16..64 rows x 4096..65536 items on 128 threads) -- and selects the K best of its
    n_chunks x K candidates at the end.  The [2048, n_items] block of the reference (1.6 GB at config 3) is never materialised:
    at 200 000 items even a 64-row slab of it (51 MB) streams through DRAM six times.  Same lists up to the order of exactly
    equal scores."""
    from concurrent.futures import ThreadPoolExecutor
    import os
    threads = threads or os.cpu_count() or 1
    pool = _POOLS.get(threads)
    if pool is None:
        pool = _POOLS[threads] = ThreadPoolExecutor(max_workers=threads)
    n, n_items = users.numel(), I.shape[0]
    bounds = torch.searchsorted(coo_rows, torch.arange(0, n + slab, slab))
    out = torch.empty((n, K), dtype=torch.int64)
    condition = rec_type != "main_branch"

    def piece(s):
        lo, hi = s * slab, min(n, (s + 1) * slab)
        a, b = int(bounds[s]), int(bounds[s + 1])
        cols, order = torch.sort(coo_cols[a:b])
        rows = (coo_rows[a:b] - lo)[order]
        cb = torch.searchsorted(cols, torch.arange(0, n_items + chunk, chunk))
        Us = U.index_select(0, users[lo:hi])
        vals, idxs = [], []
        for c in range((n_items + chunk - 1) // chunk):
            c0, c1 = c * chunk, min(n_items, (c + 1) * chunk)
            R = Us @ I[c0:c1].t()
            if condition:
                R = (F.elu(R) + 1.0) * pop[c0:c1].unsqueeze(0)
            x, y = int(cb[c]), int(cb[c + 1])
            if y > x:
                R[rows[x:y], cols[x:y] - c0] = float("-inf")
            v, i = torch.topk(R, min(K, c1 - c0), dim=1, sorted=False)
            vals.append(v)
            idxs.append(i + c0)
        v, i = torch.cat(vals, 1), torch.cat(idxs, 1)
        out[lo:hi] = torch.gather(i, 1, torch.topk(v, K, dim=1, sorted=True).indices)
    list(pool.map(piece, range((n + slab - 1) // slab)))
    return out


def eval_block_native(U, I, pop, users, coo_rows, coo_cols, K=50, rec_type="condition"):
    """The block through oracle/pda_cpu_port.c: the path as one would write it in C for host cores -- fused (no rating matrix),
    AVX2 + FMA dot products on 32 users x 4 items cache blocks, the head bounded before the exponential, a K-entry heap per user,
    users over all OpenMP threads.  The fair native CPU figure beside the torch restatement of the TF op sequence."""
    import numpy as np
    from . import c_oracle
    n = users.numel()
    indptr = torch.searchsorted(coo_rows.contiguous(), torch.arange(0, n + 1)).numpy().astype(np.int64)
    idx, _ = c_oracle.cpu_port_score_topk(U.numpy(), I.numpy(), users.numpy().astype(np.int32), K, 0 if rec_type == "main_branch" else 1,
                                          None if rec_type == "main_branch" else pop.numpy(), indptr, coo_cols.numpy().astype(np.int32))
    return idx


def eval_block_reference_topk(U, I, pop, users, coo_rows, coo_cols, K=50, rec_type="condition", threads=None):
    """Scores, head and mask as eval_block; the selection by the REFERENCE'S OWN native top-K (util/cython/include/arg_topk.h:29
    arg_top_k_2d, compiled where it lies into oracle/_ref by oracle/Makefile) with its own thread pool.  None when oracle/_ref
    is absent."""
    from . import c_oracle
    if c_oracle.ref_lib() is None:
        return None
    R = U.index_select(0, users) @ I.t()
    if rec_type != "main_branch":
        R = (F.elu(R) + 1.0) * pop.unsqueeze(0)
    R[coo_rows, coo_cols] = float("-inf")
    return c_oracle.ref_arg_topk(R.numpy(), K, threads=threads or __import__("os").cpu_count())


def time_eval(U, I, pop, users_blocks, coo_blocks, K=50, rec_type="condition", budget_s=20.0, warmups=3, reps=10, block_fn=None):
    """BASELINE.md section 3 protocol: `warmups` untimed reference blocks, then the MEDIAN block time of up to `reps` timed
    blocks (fewer when `budget_s` of wall time is used up first; at least one).  Returns (users/s, users timed)."""
    import statistics
    fn = block_fn or eval_block
    if fn is eval_block_slabbed or fn is eval_block_blocked:      # the slabs are the parallelism: one torch thread inside every piece
        prev = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            return time_eval(U, I, pop, users_blocks, coo_blocks, K, rec_type, budget_s, warmups, reps,
                             block_fn=lambda *a: fn(*a, threads=prev))
        finally:
            torch.set_num_threads(prev)
    t_start = time.perf_counter()
    blocks = list(zip(users_blocks, coo_blocks))
    for users, (rows, cols) in blocks[:warmups]:
        fn(U, I, pop, users, rows, cols, K, rec_type)
        if time.perf_counter() - t_start > budget_s * 0.4:
            break
    times, done = [], 0
    for users, (rows, cols) in blocks[warmups:warmups + reps]:
        t0 = time.perf_counter()
        fn(U, I, pop, users, rows, cols, K, rec_type)
        times.append((time.perf_counter() - t0) / users.numel())
        done += users.numel()
        if time.perf_counter() - t_start > budget_s:
            break
    return 1.0 / statistics.median(times), done


class AdamDenseDecay:
    """TF-1.14 AdamOptimizer sparse apply: decay m,v and update var on EVERY row each step [TF-ext]."""

    def __init__(self, params, lr, b1=0.9, b2=0.999, eps=1e-8):
        self.params, self.lr, self.b1, self.b2, self.eps, self.t = params, lr, b1, b2, eps, 0
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]

    @torch.no_grad()
    def step(self):
        self.t += 1
        lr_t = self.lr * (1 - self.b2 ** self.t) ** 0.5 / (1 - self.b1 ** self.t)
        for p, m, v in zip(self.params, self.m, self.v):
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            p.sub_(lr_t * m / (v.sqrt() + self.eps))
            p.grad = None


def train_step(U, I, opt, users, pos, neg, pos_pop, neg_pop, regs, batch_size):
    ue, pe, ne = U.index_select(0, users), I.index_select(0, pos), I.index_select(0, neg)
    ps, ns = (ue * pe).sum(1), (ue * ne).sum(1)
    if pos_pop is not None:
        ps, ns = (F.elu(ps) + 1) * pos_pop, (F.elu(ns) + 1) * neg_pop
    mf = -torch.log(torch.sigmoid(ps - ns) + 1e-10).mean()
    reg = regs * 0.5 * ((ue ** 2).sum() + (pe ** 2).sum() + (ne ** 2).sum()) / batch_size
    (mf + reg).backward()
    opt.step()
    return float(mf + reg), float(mf), float(reg)


def time_train(U, I, batches, regs, batch_size, lr, budget_s=10.0):
    """`batches`: list of (users, pos, neg, pos_pop, neg_pop) CPU tensors.  Returns (triplets/s, n_steps)."""
    U = U.clone().requires_grad_(True)
    I = I.clone().requires_grad_(True)
    opt = AdamDenseDecay([U, I], lr)
    n, t0 = 0, time.perf_counter()
    while True:
        for b in batches:
            train_step(U, I, opt, *b, regs, batch_size)
            n += 1
            if time.perf_counter() - t0 > budget_s:
                dt = time.perf_counter() - t0
                return n * batch_size / dt, n
