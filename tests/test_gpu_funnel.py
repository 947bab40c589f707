"""The funnel (pda_score_topk7_*, pda_v7_funnel.h) against the oracle: the raw head's top-K through fixed-threshold emitting sweeps, bound-keyed pools
and ONE exact rescoring at the end must return the oracle's lists bit for bit -- on ragged blocks, with and without the train-item mask, when
bets are lost and lists overflow (the in-call exact fallback), on exact ties, on a user without enough unmasked items, and on rows built to hit
the WORST case of the bf16 rounding bound (the advisor's round-4 finding: the filters assumed half the true unit roundoff).
Reference: MF/model_api.py:62 + tf.nn.top_k behind the -inf mask, MF/train_new_api.py:594-612."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import c_oracle

pytestmark = pytest.mark.gpu
ENV = ("PDA_SCORE_IMPL", "PDA_SCORE_KERNEL", "PDA_SCORE_LISTS", "PDA_SCORE_PRUNE", "PDA_SCORE_FUNNEL")


@pytest.fixture(autouse=True)
def clean_env(monkeypatch):
    for k in ENV:
        monkeypatch.delenv(k, raising=False)


def csr(rows):
    ip = np.zeros(len(rows) + 1, dtype=np.int64)
    ip[1:] = np.cumsum([len(r) for r in rows])
    ix = np.concatenate([np.sort(np.asarray(r, dtype=np.int32)) for r in rows]) if len(rows) else np.zeros(0, np.int32)
    return ip, ix.astype(np.int32)


def funnel(ops, U, I, users, K, hist, monkeypatch):
    monkeypatch.setenv("PDA_SCORE_FUNNEL", "1")
    st = {}
    keys = ops.score_topk_keys(U, I, users, K, ops.HEAD_RAW, None, hist, stats=st)
    assert keys.shape == (1, users.numel(), K)
    ident = ops.kernel_identity(st["kernel_id"][0])
    assert ident["generation"] == 4 and ident["geometry"] == "funnel" and ident["head"] == 0, ident
    assert int(st["error"][0]) == 0
    return ops.unpack_keys(keys[0]), int(st["fallback_rows"][0])


def tune(fail_p=0.0, growth=0, cap_e=0, first_tiles=0):
    from pda_amd import _lib
    L = _lib.load()
    L.pda_debug_funnel_tune.restype, L.pda_debug_funnel_tune.argtypes = C.c_int, [C.c_double, C.c_int, C.c_int, C.c_int]
    assert L.pda_debug_funnel_tune(fail_p, growth, cap_e, first_tiles) == 0


DEFAULTS = dict(fail_p=1e-6, growth=4, cap_e=64, first_tiles=4)


def maxima(on):
    from pda_amd import _lib
    L = _lib.load()
    L.pda_debug_funnel_maxima.restype, L.pda_debug_funnel_maxima.argtypes = C.c_int, [C.c_int]
    assert L.pda_debug_funnel_maxima(int(on)) == 0


def make(rng, nU, nI, d, scale=0.1):
    U = (rng.standard_normal((nU, d)) * scale).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * scale * (0.5 + rng.random((nI, 1)))).astype(np.float32)
    return U, I


@pytest.mark.parametrize("d,K,with_hist,bf16", [(128, 50, True, False), (64, 50, True, False), (128, 7, False, False), (64, 54, False, False), (128, 50, True, True),
                                                   (256, 50, True, False), (256, 20, False, True)])
def test_funnel_equals_the_oracle(dev, monkeypatch, d, K, with_hist, bf16):
    """Ragged block (4 200 users: four 1 024-user tiles and a rest -- eight 512-user tiles and a rest at d = 256; 9 000 items: 140 tiles and a rest),
    random user ids, item splits."""
    from pda_amd import ops
    rng = np.random.default_rng(100 + d + K)
    nU, nI, nu = 5000, 9000, 4200
    U, I = make(rng, nU, nI, d)
    users = rng.permutation(nU)[:nu].astype(np.int32)
    rows = [rng.choice(nI, rng.integers(0, 70), replace=False) for _ in range(nU)]
    ip, ix = csr(rows)
    Ut, It = torch.from_numpy(U).to(dev), torch.from_numpy(I).to(dev)
    if bf16:
        Ut, It = Ut.bfloat16(), It.bfloat16()
        U, I = Ut.float().cpu().numpy(), It.float().cpu().numpy()        # (the scores of bf16 tables are defined on the widened values)
    hist = ops.HistoryCSR(torch.from_numpy(ip).to(dev), torch.from_numpy(ix).to(dev), by_user=True) if with_hist else None
    (idx, val), nfb = funnel(ops, Ut, It, torch.from_numpy(users).to(dev), K, hist, monkeypatch)
    bip, bix = csr([rows[u] for u in users])
    ridx, rval = c_oracle.score_topk(U[users], I, np.arange(nu, dtype=np.int32), K, 0, None, bip if with_hist else None, bix if with_hist else None, order=1)
    np.testing.assert_array_equal(val, rval)
    np.testing.assert_array_equal(idx, ridx)
    assert nfb <= nu // 100


@pytest.mark.parametrize("force_failures", [False, True])
def test_funnel_with_a_mask_by_block_row(dev, monkeypatch, force_failures):
    """The reference's per-block mask (a COO triple whose rows are the block's rows, MF/train_new_api.py:736,791 -> HistoryCSR by_user=False): the funnel serves it;
    rows that fail (forced here: bold bets, two-entry lists) are swept again IN PLACE by generation 4 -- the oracle's lists either way."""
    from pda_amd import ops
    rng = np.random.default_rng(31)
    nU, nI, nu, d, K = 4000, 30000, 2048, 64, 50
    U, I = make(rng, nU, nI, d)
    users = rng.permutation(nU)[:nu].astype(np.int32)
    rows = [rng.choice(nI, rng.integers(0, 90), replace=False) for _ in range(nu)]          # row r of the BLOCK
    bip, bix = csr(rows)
    hist = ops.HistoryCSR(torch.from_numpy(bip).to(dev), torch.from_numpy(bix).to(dev), by_user=False)
    if force_failures:
        tune(fail_p=0.5, cap_e=2)
    try:
        assert ops.score_plan(nu, nI, d, K, ops.HEAD_RAW, None, hist=hist)["kernel"] == "funnel"
        st = {}
        keys = ops.score_topk_keys(torch.from_numpy(U).to(dev), torch.from_numpy(I).to(dev), torch.from_numpy(users).to(dev), K, ops.HEAD_RAW, None, hist, stats=st)
        ident = ops.kernel_identity(st["kernel_id"][0])
        assert ident["geometry"] == "funnel" and int(st["error"][0]) == 0
        nfb = int(st["fallback_rows"][0])
    finally:
        tune(**DEFAULTS)
    assert (nfb > nu // 10) if force_failures else (nfb <= nu // 100)
    idx, val = ops.unpack_keys(keys[0])
    ridx, rval = c_oracle.score_topk(U[users], I, np.arange(nu, dtype=np.int32), K, 0, None, bip, bix, order=1)
    np.testing.assert_array_equal(val, rval)
    np.testing.assert_array_equal(idx, ridx)


def test_funnel_refuses_a_prep_whose_image_is_bf16(dev):
    """pda_score_topk7_* wants pda_item_prep7_* (fp16 image: its sweeps run v_mfma_f32_16x16x32_f16).  Handed a pda_item_prep4_* prep it must not form products of
    reinterpreted bits: error word 7, nothing scored, every row through the exact fallback -- and the oracle's lists all the same."""
    from pda_amd import _lib, ops
    from pda_amd._lib import ptr, stream_ptr
    rng = np.random.default_rng(5)
    nU, nI, d, K = 1500, 9000, 128, 50
    U, I = make(rng, nU, nI, d)
    Ut, It = torch.from_numpy(U).to(dev), torch.from_numpy(I).to(dev)
    users = torch.arange(nU, dtype=torch.int32, device=dev)
    lib = _lib.load()
    prep = ops.item_prep4(It, None, ops.funnel_order(It))
    ws = torch.zeros(lib.pda_score_topk7_workspace_bytes(nU, nI, d), dtype=torch.uint8, device=dev)
    keys = torch.empty((nU, K), dtype=torch.int64, device=dev)
    rc = lib.pda_score_topk7_f32(ptr(Ut), ptr(It), ptr(prep), ptr(users), nU, 0, nI, d, None, None, 0, K, ops.HEAD_RAW, ptr(keys), ptr(ws), stream_ptr())
    torch.cuda.synchronize()
    assert rc == 0
    assert int(ws[0:4].view(torch.int32)[0]) == 7 and int(ws[24:28].view(torch.int32)[0]) == nU
    idx, val = ops.unpack_keys(keys)
    ridx, rval = c_oracle.score_topk(U, I, np.arange(nU, dtype=np.int32), K, 0, None, None, None, order=1)
    np.testing.assert_array_equal(val, rval)
    np.testing.assert_array_equal(idx, ridx)


def test_funnel_on_item_shards_merges_to_the_oracle(dev, monkeypatch):
    """Item-sharded evaluation (one shard per rank): the funnel on two ragged shards of the catalogue with GLOBAL item ids in the train rows (item_offset),
    the two partial lists merged by pda_topk_merge -- the oracle's lists over the whole catalogue, bit for bit."""
    from pda_amd import ops
    rng = np.random.default_rng(77)
    nU, nI, nu, d, K = 3000, 40000, 2500, 128, 50
    U, I = make(rng, nU, nI, d)
    users = rng.permutation(nU)[:nu].astype(np.int32)
    rows = [rng.choice(nI, rng.integers(0, 70), replace=False) for _ in range(nU)]
    ip, ix = csr(rows)
    Ut, It = torch.from_numpy(U).to(dev), torch.from_numpy(I).to(dev)
    hist = ops.HistoryCSR(torch.from_numpy(ip).to(dev), torch.from_numpy(ix).to(dev), by_user=True)
    ut = torch.from_numpy(users).to(dev)
    monkeypatch.setenv("PDA_SCORE_FUNNEL", "1")
    parts = []
    for lo, hi in ((0, 17001), (17001, nI)):
        st = {}
        k = ops.score_topk_keys(Ut, It[lo:hi].contiguous(), ut, K, ops.HEAD_RAW, None, hist, item_offset=lo, stats=st)
        assert ops.kernel_identity(st["kernel_id"][0])["geometry"] == "funnel" and int(st["error"][0]) == 0
        parts.append(k)
    idx, val = ops.unpack_keys(ops.topk_merge(torch.cat(parts, dim=0), want="keys"))
    bip, bix = csr([rows[u] for u in users])
    ridx, rval = c_oracle.score_topk(U[users], I, np.arange(nu, dtype=np.int32), K, 0, None, bip, bix, order=1)
    np.testing.assert_array_equal(val, rval)
    np.testing.assert_array_equal(idx, ridx)


@pytest.mark.parametrize("wl", ["c1", "c2"])
def test_configs_1_and_2_take_the_funnel_unforced(dev, wl):
    """BASELINE configs 1 and 2 (47 890 x 26 047 and 50 000 x 20 000, d = 64; config 1's only model head is the raw one): NO PDA_* variable set, all users in one
    block -- the library's plan sends the raw head to the funnel (identity word; four item splits), the keys equal generation 4's, and a 256-user sample equals
    the oracle on the workload's real train rows."""
    from pda_amd import ops, synthetic
    W = synthetic.make_workload(wl, dev)
    hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
    users = torch.arange(W.n_users, dtype=torch.int32, device=dev)
    assert ops.score_plan(W.n_users, W.n_items, W.d, 50, ops.HEAD_RAW, None, hist=hist)["kernel"] == "funnel"
    st = {}
    keys = ops.topk_merge(ops.score_topk_keys(W.U, W.I, users, 50, ops.HEAD_RAW, None, hist, stats=st), want="keys")
    ident = ops.kernel_identity(st["kernel_id"][0])
    assert ident["generation"] == 4 and ident["geometry"] == "funnel" and ident["d"] == 64 and int(st["error"][0]) == 0, ident
    assert int(st["fallback_rows"][0]) <= W.n_users // 1000
    import os
    os.environ["PDA_SCORE_FUNNEL"] = "0"
    try:
        g4 = ops.topk_merge(ops.score_topk_keys(W.U, W.I, users, 50, ops.HEAD_RAW, None, hist), want="keys")
    finally:
        del os.environ["PDA_SCORE_FUNNEL"]
    assert torch.equal(keys, g4)
    n = 256
    ip, ix = W.hist_indptr.cpu().numpy(), W.hist_indices.cpu().numpy()
    bip, bix = csr([ix[ip[u]:ip[u + 1]] for u in range(n)])
    ridx, rval = c_oracle.score_topk(W.U[:n].float().cpu().numpy(), W.I.float().cpu().numpy(), np.arange(n, dtype=np.int32), 50, 0, None, bip, bix, order=1)
    idx, val = ops.unpack_keys(keys[:n])
    np.testing.assert_array_equal(val, rval)
    np.testing.assert_array_equal(idx, ridx)


@pytest.mark.parametrize("d", [128, 256])
def test_funnel_lost_bets_and_overflowing_lists_take_the_exact_fallback(dev, monkeypatch, d):
    """Thresholds that are far too bold (every second bet lost) and lists of two entries: most rows end in generation 4's exact lists inside the
    same call (seeded with the rows' tk; d = 256: generation 4's 256-user geometry) -- and the result does not move."""
    from pda_amd import ops
    rng = np.random.default_rng(7)
    nU, nI, nu, K = 3000, 8000, 2500, 50
    U, I = make(rng, nU, nI, d)
    users = np.arange(nu, dtype=np.int32)
    rows = [rng.choice(nI, rng.integers(0, 40), replace=False) for _ in range(nU)]
    ip, ix = csr(rows)
    Ut, It = torch.from_numpy(U).to(dev), torch.from_numpy(I).to(dev)
    hist = ops.HistoryCSR(torch.from_numpy(ip).to(dev), torch.from_numpy(ix).to(dev), by_user=True)
    ridx, rval = c_oracle.score_topk(U, I, users, K, 0, None, *csr([rows[u] for u in users]), order=1)
    try:
        tune(fail_p=0.5)
        (idx, val), nfb = funnel(ops, Ut, It, torch.from_numpy(users).to(dev), K, hist, monkeypatch)
        assert nfb > nu // 20                                  # the bets ARE lost
        np.testing.assert_array_equal(idx, ridx)
        np.testing.assert_array_equal(val, rval)
        tune(fail_p=1e-6, cap_e=2)
        (idx, val), nfb = funnel(ops, Ut, It, torch.from_numpy(users).to(dev), K, hist, monkeypatch)
        assert nfb > nu // 2                                   # two entries per list: nearly every row overflows
        np.testing.assert_array_equal(idx, ridx)
        np.testing.assert_array_equal(val, rval)
    finally:
        tune(**DEFAULTS)


def test_funnel_without_the_maxima_launch(dev, monkeypatch):
    """The first launch as an emitting launch against -inf (256 items, every value written, the train items masked by threshold7_kernel) instead of the
    maxima launch: the same lists."""
    from pda_amd import ops
    rng = np.random.default_rng(17)
    nU, nI, nu, d, K = 2600, 7000, 2100, 64, 50
    U, I = make(rng, nU, nI, d)
    users = np.arange(nu, dtype=np.int32)
    rows = [rng.choice(nI, rng.integers(0, 40), replace=False) for _ in range(nU)]
    rows[3] = np.arange(0, 300)                                    # (a user whose train items fill the first launch's items and more)
    ip, ix = csr(rows)
    Ut, It = torch.from_numpy(U).to(dev), torch.from_numpy(I).to(dev)
    hist = ops.HistoryCSR(torch.from_numpy(ip).to(dev), torch.from_numpy(ix).to(dev), by_user=True)
    ridx, rval = c_oracle.score_topk(U, I, users, K, 0, None, *csr([rows[u] for u in users]), order=1)
    try:
        maxima(False)
        (idx, val), nfb = funnel(ops, Ut, It, torch.from_numpy(users).to(dev), K, hist, monkeypatch)
        np.testing.assert_array_equal(idx, ridx)
        np.testing.assert_array_equal(val, rval)
    finally:
        maxima(True)
    (idx, val), _ = funnel(ops, Ut, It, torch.from_numpy(users).to(dev), K, hist, monkeypatch)
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_array_equal(val, rval)


def test_funnel_ties_zero_rows_and_short_lists(dev, monkeypatch):
    """A user with a zero row (every score 0: the K lowest unmasked ids win, tf.nn.top_k's rule), duplicated items (exact ties at the top), a user whose
    history leaves 30 items (a list shorter than K: empty slots), a user whose history is the whole catalogue."""
    from pda_amd import ops
    rng = np.random.default_rng(3)
    nU, nI, d, K = 1100, 6000, 64, 50
    U, I = make(rng, nU, nI, d)
    U[5] = 0.0
    I[100:140] = I[99]                                          # 41 identical items
    I[2000] = I[17] * 1.0
    rows = [rng.choice(nI, rng.integers(0, 30), replace=False) for _ in range(nU)]
    rows[9] = np.setdiff1d(np.arange(nI), rng.choice(nI, 30, replace=False))
    rows[11] = np.arange(nI)
    ip, ix = csr(rows)
    users = np.arange(nU, dtype=np.int32)
    Ut, It = torch.from_numpy(U).to(dev), torch.from_numpy(I).to(dev)
    hist = ops.HistoryCSR(torch.from_numpy(ip).to(dev), torch.from_numpy(ix).to(dev), by_user=True)
    (idx, val), nfb = funnel(ops, Ut, It, torch.from_numpy(users).to(dev), K, hist, monkeypatch)
    ridx, rval = c_oracle.score_topk(U, I, users, K, 0, None, ip, ix, order=1)
    fin = np.isfinite(rval)
    np.testing.assert_array_equal(val[fin], rval[fin])
    np.testing.assert_array_equal(idx[fin], ridx[fin])
    assert (idx[~fin] == -1).all() and np.isinf(val[~fin]).all()           # (the library leaves empty slots; pda_topk_merge completes them like tf.nn.top_k)
    assert fin[9].sum() == 30 and fin[11].sum() == 0
    assert nfb >= 3                                                        # the zero row, the short rows: not the funnel's to finish


def worst_case_tables(d, n_items, n_users, rng):
    """User 0 and twenty target items are built so that BOTH roundings to bf16 lose a full unit roundoff (2^-8) in the same direction on the one
    element that carries the score: s~ underestimates s by 2^-7 ||u|| ||i||, the whole worst-case bound.  200 decoys sit between the targets'
    old upper bound (2^-8 ||u|| ||i|| above s~: what rounds 1 - 4 assumed) and the targets' exact score; everything else scores ~0.  The exact top
    20 of user 0 are the targets; a filter with half the bound drops them -- IF its thresholds stand when the targets come by: 600 fillers of large
    norm and no score fill every warm-up, then the decoys (norm 2.06, low ids), 2 000 spacers, then the targets (norm 2.008, high ids), in the
    order by norm and in the order by id alike."""
    a = np.float32(1.0 + 2.0 ** -8 - 2.0 ** -13)               # bf16(a) = 1: a full unit roundoff lost
    bt = np.float32(2.0)                                       # the targets' bf16 image
    b = np.float32(bt * (1.0 + 2.0 ** -8 - 2.0 ** -13))        # ... and what they really are
    h = np.float32(2.0 ** -4)
    U = (rng.standard_normal((n_users, d)) * 0.01).astype(np.float32)
    I = (rng.standard_normal((n_items, d)) * 0.01).astype(np.float32)
    U[0] = 0.0
    U[0, 0], U[0, 1] = a, h
    fill, dec, spc, tgt = np.arange(0, 600), np.arange(600, 800), np.arange(800, 2800), np.arange(5000, 5020)
    I[fill] = 0.0
    I[fill, 3 + fill % (d - 3)] = 3.0
    I[spc] = 0.0                                               # 2 000 spacers (norm 2.03, no score): the thresholds have settled when the targets come
    I[spc, 3 + spc % (d - 3)] = 2.03
    I[tgt] = 0.0
    I[tgt, 0] = b
    I[dec] = 0.0
    I[dec, 0] = bt
    I[dec, 2] = 0.5
    # exact scores: targets a b = 2.01517; decoys a bt + h g = 2.00757 + h g with h g in [0.0012, 0.0060] (g: multiples of 2^-9, exact in bf16) -- above
    # the targets' OLD upper bound s~ + 2^-8 (1.01) ||u|| ||i|| = 2.00797 and below the targets
    I[dec, 1] = (np.float32(2.0 ** -9) * (10 + np.arange(200) // 5)).astype(np.float32)
    return U, I, np.sort(tgt)


@pytest.mark.parametrize("path", ["funnel", "v4many", "v3", "huge", "lds"])
def test_filter_bound_worst_case_rounding(dev, monkeypatch, path):
    """Every pre-filtered kernel keeps the pairs whose bf16 product underestimates the score by the full 2^-7 ||u|| ||i||."""
    from pda_amd import ops
    rng = np.random.default_rng(5)
    d, nI, nu, K = 64, 6000, 1100, 50
    U, I, tgt = worst_case_tables(d, nI, nu, rng)
    users = np.arange(nu, dtype=np.int32)
    Ut, It, ut = torch.from_numpy(U).to(dev), torch.from_numpy(I).to(dev), torch.from_numpy(users).to(dev)
    head, pop = 0, None
    if path == "funnel":
        monkeypatch.setenv("PDA_SCORE_FUNNEL", "1")
    elif path == "v3":
        monkeypatch.setenv("PDA_SCORE_KERNEL", "v3")
    elif path == "v4many":
        monkeypatch.setenv("PDA_SCORE_KERNEL", "v4")
        monkeypatch.setenv("PDA_SCORE_LISTS", "many")
    else:                                                       # the popularity head's geometries, popularity 1 everywhere: the same ranking
        monkeypatch.setenv("PDA_SCORE_KERNEL", "v4")
        monkeypatch.setenv("PDA_SCORE_LISTS", path)
        head, pop = 1, torch.ones(nI, dtype=torch.float32, device=dev)
    keys = ops.score_topk_keys(Ut, It, ut, K, head, pop, None, prune=("order" if head else None))
    idx, val = ops.unpack_keys(ops.topk_merge(keys, want="keys"))
    ridx, rval = c_oracle.score_topk(U, I, users, K, head, None if pop is None else np.ones(nI, np.float32), order=1)
    assert set(ridx[0, :20]) == set(tgt)                        # (the construction: the targets ARE user 0's best twenty)
    np.testing.assert_array_equal(idx[0], ridx[0])
    np.testing.assert_array_equal(idx[:64], ridx[:64])
    if head == 0:
        np.testing.assert_array_equal(val[:64], rval[:64])


@pytest.mark.parametrize("d,nI,nu", [(64, 5000, 1), (64, 5000, 7), (128, 5000, 64), (128, 30000, 500), (64, 30000, 1000), (256, 9000, 130)])
def test_blocks_of_few_users_take_the_funnel_unforced(dev, monkeypatch, d, nI, nu):
    """Round 6: the library's plan sends the raw head through the funnel from 4 096 items on whatever the number of users (a block below one 1 024-user tile
    fills it partly: 0.25 - 0.38 ms against generation 4's 2 - 3 ms at 200 000 items).  No PDA_* variable set: the identity word says funnel, the lists equal
    the oracle's bit for bit and generation 4's (forced) key for key."""
    from pda_amd import ops
    rng = np.random.default_rng(7 + d + nu)
    nU = 3000
    U, I = make(rng, nU, nI, d)
    users = rng.permutation(nU)[:nu].astype(np.int32)
    rows = [rng.choice(nI, rng.integers(0, 60), replace=False) for _ in range(nU)]
    ip, ix = csr(rows)
    Ut, It, ut = torch.from_numpy(U).to(dev), torch.from_numpy(I).to(dev), torch.from_numpy(users).to(dev)
    hist = ops.HistoryCSR(torch.from_numpy(ip).to(dev), torch.from_numpy(ix).to(dev), by_user=True)
    st = {}
    keys = ops.topk_merge(ops.score_topk_keys(Ut, It, ut, 50, ops.HEAD_RAW, None, hist, stats=st), want="keys")
    ident = ops.kernel_identity(st["kernel_id"][0])
    assert ident["geometry"] == "funnel" and int(st["error"][0]) == 0, ident
    gi, gv = ops.unpack_keys(keys)
    bip, bix = csr([rows[u] for u in users])
    ridx, rval = c_oracle.score_topk(U[users], I, np.arange(nu, dtype=np.int32), 50, 0, None, bip, bix, order=1)
    np.testing.assert_array_equal(gv, rval)
    np.testing.assert_array_equal(gi, ridx)
    monkeypatch.setenv("PDA_SCORE_FUNNEL", "0")
    st4 = {}
    k4 = ops.topk_merge(ops.score_topk_keys(Ut, It, ut, 50, ops.HEAD_RAW, None, hist, stats=st4), want="keys")
    if "kernel_id" in st4:                                        # (d = 256: generation 3 serves the forced-off call and writes no identity word)
        assert ops.kernel_identity(st4["kernel_id"][0]).get("geometry") != "funnel"
    assert torch.equal(k4, keys)
