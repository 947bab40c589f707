"""CPU suite, part 4: the N>1 evaluation path under gloo, world_size 2.  The HIP entry points cannot run here, so
the orchestration (shard ranges, local merge, ONE all-gather, final merge, set_popularity) is exercised with
test doubles that answer through the CPU oracle; the result must equal the unsharded oracle bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import c_oracle
from oracle import pda_oracle as po


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pack(vals, idxs):
    hi = vals.view(np.uint32).astype(np.uint64)
    ordb = np.where(hi & np.uint64(0x80000000), (~hi) & np.uint64(0xFFFFFFFF), hi | np.uint64(0x80000000))
    return ((ordb << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - idxs.astype(np.uint64))).view(np.int64)


def _unpack(keys):
    k = keys.view(np.uint64)
    hi = (k >> np.uint64(32)).astype(np.uint32)
    bits = np.where(hi & np.uint32(0x80000000), hi ^ np.uint32(0x80000000), ~hi).astype(np.uint32)
    return bits.view(np.float32), (np.uint32(0xFFFFFFFF) - (k & np.uint64(0xFFFFFFFF)).astype(np.uint32)).astype(np.int32)


def score_double(U, I_shard, users, K, head, pop_shard, hist, item_offset, n_splits):
    """Stand-in for ops.score_topk_keys: same contract ([n_splits, Bu, K] packed keys), computed by the oracle."""
    bip = bix = None
    if hist is not None:                      # hist = CSR by user id (replicated); the oracle wants block rows
        ip, ix = hist
        u = users.numpy()
        bip = np.zeros(len(u) + 1, np.int64)
        bip[1:] = np.cumsum(ip[u + 1] - ip[u])
        bix = np.concatenate([ix[ip[x]:ip[x + 1]] for x in u]).astype(np.int32)
    pop_full = None
    if pop_shard is not None:
        pop_full = np.zeros(item_offset + I_shard.shape[0], np.float32)
        pop_full[item_offset:] = pop_shard.numpy()
    I_full = np.zeros((item_offset + I_shard.shape[0], I_shard.shape[1]), np.float32)
    I_full[item_offset:] = I_shard.numpy()
    idx, val = c_oracle.score_topk(U.numpy(), I_full, users.numpy(), K, head, pop_full, bip, bix, item_offset=item_offset,
                                   n_items_local=I_shard.shape[0], order=1)
    return torch.from_numpy(_pack(val + np.float32(0), idx))[None]


def merge_double(keys, users, hist, want="idx_val"):
    v, i = _unpack(keys.numpy())
    mi, mv = po.merge_partial_topk(v, i, keys.shape[2])
    if want == "keys":
        return torch.from_numpy(_pack(mv, mi))
    return torch.from_numpy(mi), torch.from_numpy(mv)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pda_amd.dist import ItemShardedTopK
    rng = np.random.default_rng(4)                                  # same data on every rank (replicated inputs)
    nU, nI, d, K = 60, 777, 32, 50
    U = (rng.standard_normal((nU, d)) * 0.1).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * 0.1).astype(np.float32)
    pop = (rng.uniform(0, 1, nI) ** 0.22).astype(np.float32)
    rows = [np.sort(rng.integers(0, nI, rng.integers(0, 25))).astype(np.int32) for _ in range(nU)]
    ip = np.zeros(nU + 1, np.int64)
    ip[1:] = np.cumsum([len(r) for r in rows])
    ix = np.concatenate(rows)
    ev = ItemShardedTopK.from_full_tables(torch.from_numpy(U), torch.from_numpy(I), torch.from_numpy(pop), rank, world,
                                          score_fn=score_double, merge_fn=merge_double)
    assert ev.I_shard.shape[0] < nI and ev.item_offset == (0 if rank == 0 else ev.item_offset)
    blocks = [torch.arange(0, 32, dtype=torch.int32), torch.arange(32, 60, dtype=torch.int32)]
    out = []
    for users, (idx, val) in zip(blocks, ev.topk_blocks(blocks, K, 1, (ip, ix))):
        u = users.numpy()
        bip = np.zeros(len(u) + 1, np.int64)
        bip[1:] = np.cumsum([len(rows[x]) for x in u])
        bix = np.concatenate([rows[x] for x in u])
        ridx, rval = c_oracle.score_topk(U, I, u, K, 1, pop, bip, bix, order=1)
        out.append(bool(np.array_equal(idx.numpy(), ridx) and np.array_equal(val.numpy(), rval)))
    # sharded mode: all-to-all, this rank merges only its slice of the users
    u = blocks[0].numpy()
    bip = np.zeros(len(u) + 1, np.int64)
    bip[1:] = np.cumsum([len(rows[x]) for x in u])
    bix = np.concatenate([rows[x] for x in u])
    ridx, rval = c_oracle.score_topk(U, I, u, K, 1, pop, bip, bix, order=1)
    lo, hi = ev.user_slice(len(u))
    sidx, sval = ev.topk_sharded(blocks[0], K, 1, (ip, ix))
    out.append(bool(sidx.shape[0] == hi - lo and np.array_equal(sidx.numpy(), ridx[lo:hi]) and np.array_equal(sval.numpy(), rval[lo:hi])))
    # a new popularity vector is re-sliced per shard (evaluation.set_testing_popularity)
    pop2 = (pop * 0.5).astype(np.float32)
    ev.set_popularity(torch.from_numpy(pop2))
    idx, val = ev.topk(blocks[0], K, 1, None)
    ridx, rval = c_oracle.score_topk(U, I, blocks[0].numpy(), K, 1, pop2, order=1)
    out.append(bool(np.array_equal(idx.numpy(), ridx)))
    q.put((rank, out, ev.item_offset, ev.I_shard.shape[0]))
    dist.barrier()
    dist.destroy_process_group()


def test_item_sharded_eval_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    assert all(all(r[1]) for r in res), res
    (o0, n0), (o1, n1) = (res[0][2], res[0][3]), (res[1][2], res[1][3])
    assert o0 == 0 and o1 == n0 and n0 + n1 == 777 and n0 % 32 == 0


def _worker_2d(rank, world, port, q):
    """Four ranks as 2 user groups x 2 item shards (pda_amd.dist.grid_layout): a group scores ITS half of the users of a
    block against its two item shards; the union over groups and ranks is the unsharded result."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pda_amd.dist import ItemShardedTopK, make_item_group
    rng = np.random.default_rng(5)
    nU, nI, d, K = 64, 500, 32, 20
    U = (rng.standard_normal((nU, d)) * 0.1).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * 0.1).astype(np.float32)
    pop = (rng.uniform(0, 1, nI) ** 0.22).astype(np.float32)
    g, r, gsize, pg = make_item_group(rank, world, 2)
    ev = ItemShardedTopK.from_full_tables(torch.from_numpy(U), torch.from_numpy(I), torch.from_numpy(pop), r, gsize, group=pg,
                                          score_fn=score_double, merge_fn=merge_double)
    users = torch.arange(g * 32, (g + 1) * 32, dtype=torch.int32)          # this group's half of the 64-user block
    lo, hi = ev.user_slice(users.numel())
    sidx, sval = ev.topk_sharded(users, K, 1, None)
    ridx, rval = c_oracle.score_topk(U, I, users.numpy(), K, 1, pop, order=1)
    ok = bool(np.array_equal(sidx.numpy(), ridx[lo:hi]) and np.array_equal(sval.numpy(), rval[lo:hi]))
    q.put((rank, ok, g, r, gsize, ev.item_offset, int(users[lo]), int(users[hi - 1])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_user_groups_of_two_item_shards_gloo():
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_2d, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    assert [(r[2], r[3], r[4]) for r in res] == [(0, 0, 2), (0, 1, 2), (1, 0, 2), (1, 1, 2)]
    assert res[0][5] == 0 and res[1][5] > 0 and res[2][5] == 0            # item offsets repeat per group
    covered = sorted((r[6], r[7]) for r in res)                            # the four user slices tile the block
    assert covered == [(0, 15), (16, 31), (32, 47), (48, 63)]


# ---- item-parallel training step (ItemShardedBPR) -----------------------------------------------------------------

def _global_batch(rng, R, nU, nI, Bl):
    per = nI // R
    users = rng.permutation(nU)[:R * Bl].astype(np.int32)
    pos = np.concatenate([rng.integers(r * per, (r + 1) * per, Bl) for r in range(R)]).astype(np.int32)
    neg = np.concatenate([rng.integers(r * per, (r + 1) * per, Bl) for r in range(R)]).astype(np.int32)
    pp = (rng.uniform(0, 1, R * Bl) ** 0.22).astype(np.float32)
    pn = (rng.uniform(0, 1, R * Bl) ** 0.22).astype(np.float32)
    return users, pos, neg, pp, pn


def step_double(U, I_shard, item_offset, users, pos, neg, pos_pop, neg_pop, *, regs, reg_div, mean_div, lr, g_user, loss_acc):
    """Stand-in for ops.bpr_step_shard (same contract), answered by the float64 oracle: the per-triplet gradients of a
    sub-batch inside a global batch of `mean_div` triplets are those of a batch padded to that length."""
    Bl, Bg = users.numel(), int(mean_div)
    I_full = np.zeros((item_offset + I_shard.shape[0], I_shard.shape[1]), np.float32)
    I_full[item_offset:] = I_shard.numpy()
    fw = po.bpr_forward(U.numpy(), I_full, users.numpy(), pos.numpy(), neg.numpy(), pos_pop.numpy(), neg_pop.numpy())
    scale = Bl / Bg                                        # oracle means over len(batch) = Bl; the global step over Bg
    due, dpe, dne = po.bpr_grads(fw, 0.0, reg_div, pos_pop.numpy(), neg_pop.numpy())
    c = regs / reg_div
    due, dpe, dne = due * scale + c * fw["ue"], dpe * scale + c * fw["pe"], dne * scale + c * fw["ne"]
    g_user.copy_(torch.from_numpy(due.astype(np.float32)))
    upd = I_full.astype(np.float64)
    np.subtract.at(upd, pos.numpy(), lr * dpe)
    np.subtract.at(upd, neg.numpy(), lr * dne)
    I_shard.copy_(torch.from_numpy(upd[item_offset:].astype(np.float32)))
    loss, mf, reg = po.bpr_loss(fw, regs, reg_div)
    loss_acc += torch.tensor([mf * scale + reg, mf * scale, reg], dtype=torch.float32)


def apply_double(U, users_all, g, lr):
    U.index_add_(0, users_all.long(), g.contiguous(), alpha=-lr)


def _train_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pda_amd.dist import ItemShardedBPR
    rng = np.random.default_rng(11)                                 # same data on every rank
    nU, nI, d, Bl, regs, lr = 300, 128, 32, 64, 1e-2, 0.5
    U = (rng.standard_normal((nU, d)) * 0.2).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * 0.2).astype(np.float32)
    per, Bg = nI // world, world * Bl
    Ut = torch.from_numpy(U.copy())
    shard = torch.from_numpy(I[rank * per:(rank + 1) * per].copy())
    tr = ItemShardedBPR(Ut, shard, rank * per, regs=regs, lr=lr, global_batch=Bg, rank=rank, world=world,
                        step_fn=step_double, apply_fn=apply_double)
    Uref, Iref, ok = U, I, []
    for step in range(2):
        users, pos, neg, pp, pn = _global_batch(rng, world, nU, nI, Bl)
        sl = slice(rank * Bl, (rank + 1) * Bl)
        loss = tr.step(*(torch.from_numpy(x[sl].copy()) for x in (users, pos, neg, pp, pn)))
        Uref, Iref, _, ref_loss = po.train_step(Uref, Iref, users, pos, neg, pp, pn, regs, Bg, lr, optimizer="sgd")
        ok.append(bool(np.allclose(loss.numpy(), ref_loss, atol=1e-5)))
        ok.append(bool(np.allclose(Ut.numpy(), Uref, atol=1e-5)))                     # replicas of U agree with the oracle
        ok.append(bool(np.allclose(shard.numpy(), Iref[rank * per:(rank + 1) * per], atol=1e-5)))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_item_parallel_training_two_ranks_gloo():
    """ItemShardedBPR orchestration (packed exchange buffer, the ONE all-gather, apply, loss shares) under gloo with
    oracle-backed doubles: two global steps must reproduce the oracle's SGD steps on the concatenated batches."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert all(all(r[1]) for r in res), res


def _warm_bounds(U, I_shard, users, K, head, pop_shard, hist, item_offset, n_splits, seed_shards):
    """Warm-up lists of the shard's first 16 items and the three bounds of pda_topk_seed_bounds."""
    warm = score_double(U, I_shard[:16], users, K, head, None if pop_shard is None else pop_shard[:16], hist, item_offset, n_splits)
    wv, wi = _unpack(warm.numpy()[0])
    wv = np.where(wi >= 0, wv, -np.inf).astype(np.float32)
    m = -(-K // seed_shards)
    return wv, torch.from_numpy(np.stack([wv[:, K - 1], wv[:, m - 1], -wv[:, m - 1]]))


def _thresholds(bounds, n_thr):
    b = bounds.numpy()
    lo, hi = np.maximum(b[0], -b[2]), b[1]
    is_open = np.isfinite(lo) & np.isfinite(hi) & (hi > lo)
    with np.errstate(invalid="ignore"):
        thr = np.stack([np.where(is_open, lo + (hi - lo) * np.float32((j + 1) / (n_thr + 1)), lo) for j in range(n_thr)]) if n_thr else np.zeros((0, len(lo)), np.float32)
    return lo.astype(np.float32), thr.astype(np.float32)


class SeededDouble:
    """ops.SeededCall on the CPU"""


def seeded_begin_double(U, I_shard, users, K, head, pop_shard, hist, item_offset=0, n_splits=0, seed_shards=1, prune=True, stats=None):
    from pda_amd import ops
    c = SeededDouble()
    c.full = score_double(U, I_shard, users, K, head, pop_shard, hist, item_offset, n_splits)
    c.wv, c.bounds = _warm_bounds(U, I_shard, users, K, head, pop_shard, hist, item_offset, n_splits, seed_shards)
    c.n_thr, c.K, c.counts = ops.seed_thresholds(seed_shards), K, None
    return c


def seeded_counts_double(c):
    if c.n_thr <= 0:
        return None
    _, thr = _thresholds(c.bounds, c.n_thr)
    c.counts = torch.from_numpy(np.stack([(c.wv >= t[:, None]).sum(1) for t in thr]).astype(np.int32))
    return c.counts


def seeded_finish_double(c):
    lo, thr = _thresholds(c.bounds, c.n_thr if c.counts is not None else 0)
    seed = lo.copy()
    if c.counts is not None:
        for j in range(c.n_thr):
            seed = np.where(c.counts.numpy()[j] >= c.K, np.maximum(seed, thr[j]), seed)
    v, i = _unpack(c.full.numpy()[0])
    keys = c.full.numpy()[0].copy()
    keys[(v < seed[:, None]) | (i < 0)] = 0
    c.dropped = float((keys == 0).mean())
    return torch.from_numpy(keys)[None]


def score_double_seeded(U, I_shard, users, K, head, pop_shard, hist, item_offset, n_splits, seed_reduce=None, seed_shards=1, prune=None,
                        seed_sum=None):
    """The seeded contract of ops.score_topk_keys on the CPU: warm-up lists of the shard's first 16 items -> the three bounds
    (values at rank K, at rank ceil(K / R), minus the latter) -> seed_reduce (ONE MAX over the shards, in place) -> the counts at
    ops.seed_thresholds(R) common thresholds -> seed_sum (ONE SUM) -> the shard's list WITHOUT the entries below the seed (empty
    slots = key 0)."""
    if seed_reduce is None:
        return score_double(U, I_shard, users, K, head, pop_shard, hist, item_offset, n_splits)
    c = seeded_begin_double(U, I_shard, users, K, head, pop_shard, hist, item_offset, n_splits, seed_shards)
    seed_reduce(c.bounds)
    cnt = seeded_counts_double(c) if seed_sum is not None else None
    if cnt is not None:
        seed_sum(cnt)
    keys = seeded_finish_double(c)
    return keys, c.dropped, (c.n_thr if cnt is not None else 0)


def _worker_seeded(rank, world, port, q):
    """Three ranks, 60 items = two 32-item tiles: rank 2 owns NOTHING and must still join the seed all-reduces.  Then the same
    users as three blocks through topk_blocks with the step-by-step doubles (the software-pipelined path of pda_amd.dist):
    identical lists, and at most three collectives per user block."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pda_amd.dist import ItemShardedTopK
    from pda_amd import ops
    rng = np.random.default_rng(11)
    nU, nI, d, K = 48, 60, 64, 10
    U = rng.standard_normal((nU, d), dtype=np.float32) * 0.3
    I = rng.standard_normal((nI, d), dtype=np.float32) * 0.3
    pop = (rng.uniform(0, 1, nI) ** 3).astype(np.float32)
    pop[:32] += 1.0                            # shard 0 holds the popular items: its K-th value prunes shard 1's list
    dropped, thr_seen = [], []

    def fn(*a, **k):
        keys, frac, n_thr = score_double_seeded(*a, **k)
        dropped.append(frac)
        thr_seen.append(n_thr)
        return keys
    ev = ItemShardedTopK.from_full_tables(torch.from_numpy(U), torch.from_numpy(I), torch.from_numpy(pop), rank, world,
                                          score_fn=fn, merge_fn=merge_double)
    assert ev.seeded is False
    ev.seeded = True
    ev.prune = True
    users = torch.arange(nU, dtype=torch.int32)
    idx, val = ev.topk(users, K, 1, None)
    ridx, rval = c_oracle.score_topk(U, I, users.numpy(), K, 1, pop, order=1)
    ok = bool(np.array_equal(idx.numpy(), ridx) and np.array_equal(val.numpy(), rval))
    # one validation all-reduce (once), MAX, SUM (from four shards on), all-gather
    per_block_first = ev.n_collectives
    # the pipelined path: three blocks, step-by-step doubles
    ev.seeded_api = (seeded_begin_double, seeded_counts_double, seeded_finish_double)
    ev.n_collectives = 0
    blocks = [users[0:16], users[16:32], users[32:48]]
    got = list(ev.topk_blocks(blocks, K, 1, None))
    ok2 = all(np.array_equal(i.numpy(), ridx[16 * b:16 * b + 16]) and np.array_equal(v.numpy(), rval[16 * b:16 * b + 16]) for b, (i, v) in enumerate(got))
    q.put((rank, ok and ok2, ev.I_shard.shape[0], max(dropped) if dropped else -1.0, max(thr_seen) if thr_seen else -1,
           per_block_first, ev.n_collectives / 3.0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [3, 4])
def test_seeded_item_shards_with_an_empty_shard_gloo(world):
    """world 3: the plain seed (ONE all-reduce MAX); world 4: also the counts at seven common thresholds (ONE all-reduce SUM)
    -- with TWO ranks that own nothing and still have to join every collective.  At most three collectives per user block
    including the exchange of the lists."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_seeded, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    assert [r[2] for r in res] == [32, 28, 0, 0][:world]
    assert res[1][3] > 0 and res[2][3] == -1.0         # the seed did drop entries of shard 1; rank 2 never scored
    assert res[0][4] == (7 if world == 4 else 0)
    want = 3 if world == 4 else 2                      # MAX (+ SUM) + the exchange of the lists
    assert all(r[5] == want and r[6] == want for r in res), res


# ---- replicated hot items (round 4): item shards without a warm-up per rank ------------------------------------------------------
def _worker_hot(rank, world, port, q):
    """The hot_items most popular rows are replicated (ONE all-reduce per weight version) and taken out of the shards; per block a
    rank scores its slice of the users against them, the K-th values are all-gathered as the seed, every rank sweeps its cold shard
    from empty lists against the seed, and the all-to-all + merge follow: TWO collectives per block, lists equal to the oracle's.
    world 3 leaves rank 2 without items (60 items = two 32-item tiles)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pda_amd.dist import ItemShardedTopK
    rng = np.random.default_rng(21)
    nU, nI, d, K = 48, (60 if world == 3 else 330), 32, 10
    U = rng.standard_normal((nU, d), dtype=np.float32) * 0.3
    I = rng.standard_normal((nI, d), dtype=np.float32) * 0.3
    I[7] = I[40]                                    # an exact tie between a hot and a cold candidate
    pop = (rng.uniform(0.05, 1, nI) ** 2).astype(np.float32)
    pop[7] = pop[40] = 0.9
    pop[rng.choice(nI, 11, replace=False)] += 3.0   # a popularity head: the seed of most users keeps the cold shards' lists short
    rows = [np.unique(rng.integers(0, nI, rng.integers(0, 12))).astype(np.int32) for _ in range(nU)]
    rows[5] = np.arange(nI, dtype=np.int32)[: nI - 4]              # four unmasked items: seed = -inf, every list short
    ip = np.zeros(nU + 1, np.int64)
    ip[1:] = np.cumsum([len(r) for r in rows])
    ix = np.concatenate(rows)
    seen = {"sweeps": 0, "kept": 0, "slots": 0}

    def score(U_, I_, users, K_, head, pop_, hist, off, n_splits, prune=None):
        return score_double(U_, I_, users, K_, head, pop_, hist, off, n_splits)

    def sweep_seed(U_, I_, users, K_, head, pop_, hist, off, seed, prune=None):
        keys = score_double(U_, I_, users, K_, head, pop_, hist, off, 1).numpy()[0].copy()
        v, i = _unpack(keys)
        with np.errstate(invalid="ignore"):
            keys[(v < seed.numpy()[:, None]) | (i < 0) | ~np.isfinite(v)] = 0
        seen["sweeps"] += 1
        seen["kept"] += int((keys != 0).sum())
        seen["slots"] += keys.size
        return torch.from_numpy(keys)[None]

    def kth(keys, pos):
        k = keys.numpy()[0]
        v, i = _unpack(k)
        with np.errstate(invalid="ignore"):
            return torch.from_numpy(np.where((k[:, pos] != 0) & np.isfinite(v[:, pos]), v[:, pos], -np.inf).astype(np.float32))

    ev = ItemShardedTopK.from_full_tables(torch.from_numpy(U), torch.from_numpy(I), torch.from_numpy(pop), rank, world,
                                          score_fn=score, merge_fn=merge_double)
    assert ev.hot_items == 0                       # a caller's own score_fn opts in
    ev.hot_items, ev.sweep_seed_fn, ev.kth_fn, ev.prune, ev.hot_min_shards = 12, sweep_seed, kth, "order", 2
    users = torch.arange(nU, dtype=torch.int32)
    blocks = [users[0:24], users[24:48]]
    got = list(ev.topk_blocks(blocks, K, 1, (ip, ix), sharded=True))
    ok = True
    for b, (idx, val) in enumerate(got):
        u = blocks[b].numpy()
        bip = np.zeros(len(u) + 1, np.int64)
        bip[1:] = np.cumsum([len(rows[x]) for x in u])
        bix = np.concatenate([rows[x] for x in u])
        ridx, rval = c_oracle.score_topk(U, I, u, K, 1, pop, bip, bix, order=1)
        lo, hi = ev.user_slice(len(u))
        finite = np.isfinite(rval[lo:hi])
        ok = ok and idx.shape[0] == hi - lo and bool(np.array_equal(idx.numpy()[finite], ridx[lo:hi][finite]) and np.array_equal(val.numpy()[finite], rval[lo:hi][finite]))
    per_block = ev.n_collectives / len(blocks)
    # a new popularity vector: the hot set is rebuilt (one more all-reduce of the rows), results follow
    pop2 = pop[::-1].copy()
    ev.set_popularity(torch.from_numpy(pop2))
    (idx, val), = list(ev.topk_blocks([blocks[0]], K, 1, None, sharded=True))
    ridx, rval = c_oracle.score_topk(U, I, blocks[0].numpy(), K, 1, pop2, order=1)
    lo, hi = ev.user_slice(24)
    ok2 = bool(np.array_equal(idx.numpy(), ridx[lo:hi]) and np.array_equal(val.numpy(), rval[lo:hi]))
    q.put((rank, ok, ok2, per_block, ev.n_epoch_collectives, seen["kept"] / max(1, seen["slots"]), ev.I_shard.shape[0]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_replicated_hot_items_gloo(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_hot, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] for r in res), res
    assert all(r[3] == 2 for r in res), res            # the seed all-gather and the exchange of the lists
    assert all(r[4] == 2 for r in res), res            # the hot rows: once per popularity / weight version
    assert all(r[5] < 0.5 for r in res if r[6] > 0), res   # the seed kept most of the cold pairs out of the lists
    if world == 3:
        assert res[2][6] == 0                          # a rank without items follows the same collectives
