"""GPU suite: the sweep from empty lists against a seed (pda_score_topk4_phase_*, phase 4) and the replicated-hot-items scheme of
pda_amd.dist built on it -- R emulated item shards on one GPU must reproduce the unsharded lists bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def dev():
    return torch.device("cuda:0")


def _case(rng, nU, nI, d, skew):
    U = (rng.standard_normal((nU, d)) * 0.1).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * 0.1).astype(np.float32)
    pop = (((np.arange(nI) + 1.0) ** -skew)[rng.permutation(nI)] ** 0.22).astype(np.float32)
    hist = [np.unique(rng.integers(0, nI, rng.integers(0, 40))).astype(np.int32) for _ in range(nU)]
    hist[2] = np.arange(nI, dtype=np.int32)[: nI - 9]             # nine unmasked items: no seed, every list short
    I[50:60] = I[10:20]                                           # exact ties (equal rows, equal popularity)
    pop[50:60] = pop[10:20]
    ip = np.zeros(nU + 1, np.int64)
    ip[1:] = np.cumsum([len(h) for h in hist])
    return U, I, pop, ip, np.concatenate(hist).astype(np.int32)


@pytest.mark.parametrize("d,nU,nI,K,R,H,mode,bf", [(64, 1100, 9000, 50, 4, 256, "order", False), (128, 700, 5000, 20, 8, 128, "order", False),
                                                  (128, 2300, 12000, 50, 2, 256, "order", False), (64, 900, 7000, 50, 3, 256, True, False),
                                                  (256, 300, 4000, 50, 2, 64, "order", False), (256, 1200, 6000, 50, 4, 256, "order", True),
                                                  (128, 800, 5000, 50, 3, 256, True, True)])
def test_emulated_item_shards_with_replicated_hot_items(dev, monkeypatch, d, nU, nI, K, R, H, mode, bf):
    from pda_amd import ops
    from pda_amd.dist import ItemShardedTopK, shard_range
    monkeypatch.setenv("PDA_CHECK_SWEEP_ERRORS", "1")
    rng = np.random.default_rng(d + R)
    U, I, pop, ip, ix = _case(rng, nU, nI, d, skew=1.0)
    Ut, It, popt = (torch.from_numpy(x).to(dev) for x in (U, I, pop))
    if bf:                                             # config 5's table type: both tables bf16
        Ut, It = Ut.to(torch.bfloat16), It.to(torch.bfloat16)
    hist = ops.HistoryCSR(torch.from_numpy(ip).to(dev), torch.from_numpy(ix).to(dev), by_user=True)
    users = torch.arange(nU, dtype=torch.int32, device=dev)
    want = ops.topk_merge(ops.score_topk_keys(Ut, It, users, K, 1, popt, hist, prune=mode), want="keys")

    # what dist.ItemShardedTopK._hot_state / _hot_hist build, for R emulated ranks in one process
    hot_ids = torch.argsort(popt, descending=True, stable=True)[:H].sort().values
    rows = torch.repeat_interleave(torch.arange(nU, device=dev), hist.indptr[1:] - hist.indptr[:-1])
    idx = hist.indices.long()
    pos = torch.searchsorted(hot_ids, idx).clamp_(max=H - 1)
    is_hot = hot_ids[pos] == idx

    def csr(sel, local):
        ptr = torch.zeros(nU + 1, dtype=torch.int64, device=dev)
        torch.cumsum(torch.bincount(rows[sel], minlength=nU), 0, out=ptr[1:])
        return ops.HistoryCSR(ptr, local.to(torch.int32).contiguous(), by_user=True)

    hot_keys = ops.score_topk_keys(Ut, It[hot_ids].contiguous(), users, K, 1, popt[hot_ids].contiguous(), csr(is_hot, pos[is_hot]), 0, 1, prune=mode)
    seed = ops.kth_value(hot_keys, K - 1)
    assert torch.isinf(seed[2]) and seed[2] < 0
    lists = [ItemShardedTopK.remap_keys(hot_keys[0], hot_ids.to(torch.int32))]
    kept = 0
    for r in range(R):
        lo, hi = shard_range(nI, r, R)
        mine = (hot_ids >= lo) & (hot_ids < hi)
        cold_mask = torch.ones(hi - lo, dtype=torch.bool, device=dev)
        cold_mask[hot_ids[mine] - lo] = False
        cold_local = torch.nonzero(cold_mask).flatten()
        cold_index_of = torch.cumsum(cold_mask, 0) - 1
        loc = idx - lo
        is_cold = (loc >= 0) & (loc < hi - lo) & ~is_hot
        st = {}
        k = ops.sweep_from_seed(Ut, It[lo:hi][cold_local].contiguous(), users, K, 1, popt[lo:hi][cold_local].contiguous(),
                                csr(is_cold, cold_index_of[loc[is_cold]]), 0, seed, prune=mode, stats=st)
        assert int(st["error"][0]) == 0
        kept += int((k != 0).sum())
        lists.append(ItemShardedTopK.remap_keys(ops.topk_merge(k, want="keys"), (cold_local + lo).to(torch.int32)))
    got = ops.topk_merge(torch.stack(lists).contiguous(), want="keys")
    assert torch.equal(got, want), int((got != want).sum())
    # the seed did its work: the cold shards hold a fraction of a full list per user (user 2 has no seed and keeps everything it can)
    assert kept < 0.5 * R * nU * K, kept / (R * nU * K)


def test_sweep_from_seed_without_a_bound_is_the_plain_sweep(dev, monkeypatch):
    """seed = -inf everywhere: phase 4 is a sweep of the whole shard from empty lists -- the lists of a one-call sweep."""
    from pda_amd import ops
    monkeypatch.setenv("PDA_CHECK_SWEEP_ERRORS", "1")
    rng = np.random.default_rng(3)
    for d, nU, nI, K, splits in ((64, 600, 3000, 50, 1), (128, 1500, 6000, 50, 3)):
        U, I, pop, ip, ix = _case(rng, nU, nI, d, skew=0.5)
        Ut, It, popt = (torch.from_numpy(x).to(dev) for x in (U, I, pop))
        hist = ops.HistoryCSR(torch.from_numpy(ip).to(dev), torch.from_numpy(ix).to(dev), by_user=True)
        users = torch.arange(nU, dtype=torch.int32, device=dev)
        want = ops.topk_merge(ops.score_topk_keys(Ut, It, users, K, 1, popt, hist, prune="order"), want="keys")
        seed = torch.full((nU,), float("-inf"), dtype=torch.float32, device=dev)
        for mode in ("order", True):
            got = ops.topk_merge(ops.sweep_from_seed(Ut, It, users, K, 1, popt, hist, 0, seed, n_splits=splits, prune=mode), want="keys")
            assert torch.equal(got, want), (d, mode, int((got != want).sum()))


def test_remap_key_items_matches_the_torch_expression(dev):
    """pda_topk_remap_items (in place, one launch) against ItemShardedTopK.remap_keys' torch expression on the CPU: scores of both signs,
    empty slots (key 0 stays 0), local ids at both ends of the table."""
    from pda_amd import ops
    from pda_amd.dist import ItemShardedTopK
    rng = np.random.default_rng(9)
    n_gid = 1000
    gid = np.sort(rng.choice(5_000_000, n_gid, replace=False)).astype(np.int32)
    val = rng.standard_normal((64, 50)).astype(np.float32) * 3
    loc = rng.integers(0, n_gid, (64, 50)).astype(np.int64)
    loc[0, :2] = (0, n_gid - 1)
    hi = val.view(np.uint32).astype(np.uint64)
    ordb = np.where(hi & np.uint64(0x80000000), (~hi) & np.uint64(0xFFFFFFFF), hi | np.uint64(0x80000000))
    keys = ((ordb << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - loc.astype(np.uint64))).view(np.int64)
    keys[3, 40:] = 0
    keys[7] = 0
    want = ItemShardedTopK.remap_keys(torch.from_numpy(keys.copy()), torch.from_numpy(gid))
    got = ops.remap_key_items(torch.from_numpy(keys.copy()).to(dev), torch.from_numpy(gid).to(dev)).cpu()
    assert torch.equal(got, want)
    idx, v = ops.unpack_keys(got[:3])
    assert np.array_equal(idx[0, :2], gid[[0, n_gid - 1]]) and np.array_equal(v, val[:3])
    assert int((got[7] != 0).sum()) == 0 and int((got[3, 40:] != 0).sum()) == 0
