"""GPU parity at the sizes the bench times (BASELINE configs 3 and 5): every check here states WHICH kernel it ran.

The sweep kernels write an identity word into their workspace (generation, geometry, template instance; ops.kernel_identity)
and every comparison below asserts it -- the round-3 version of these tests restored PDA_SCORE_KERNEL=old behind a try/finally
and silently compared generation 3 with itself.  All environment changes go through monkeypatch; nothing is parametrised.
Reference loops: MF/train_new_api.py:780-794 (blocks), :594-612 (heads)."""
import numpy as np
import pytest
import torch

from oracle import c_oracle

pytestmark = pytest.mark.gpu

TOL = 1e-5  # north_star: "within 1e-5 fp32 on scores"
ENV = ("PDA_SCORE_IMPL", "PDA_SCORE_KERNEL", "PDA_SCORE_LISTS", "PDA_SCORE_PRUNE", "PDA_WARM_TILES", "PDA_SEED_ROUNDS", "PDA_SCORE_FUNNEL")


@pytest.fixture(autouse=True)
def clean_env(monkeypatch):
    for k in ENV:
        monkeypatch.delenv(k, raising=False)


def csr(hist_rows):
    indptr = np.zeros(len(hist_rows) + 1, dtype=np.int64)
    indptr[1:] = np.cumsum([len(h) for h in hist_rows])
    idx = np.concatenate([np.sort(h) for h in hist_rows]).astype(np.int32) if len(hist_rows) else np.zeros(0, np.int32)
    return indptr, idx


def oracle_sample_lists(W, users_t, head, n=256):
    """c_oracle.score_topk (the fp32 fmaf chain of the kernels, order=1) on the first n of `users_t` against the WHOLE
    catalogue of workload W with the users' real train rows; bf16 tables are widened (that is how their scores are defined)."""
    sub = users_t[:n].cpu().numpy()
    lo_, hi_ = W.hist_indptr[users_t[:n].long()].cpu().numpy(), W.hist_indptr[users_t[:n].long() + 1].cpu().numpy()
    rows = [W.hist_indices[int(a):int(b)].cpu().numpy() for a, b in zip(lo_, hi_)]      # (row by row: the CSR of config 5 has 500 M entries)
    bip, bix = csr(rows)
    Uw, Iw = W.U[users_t[:n].long()].float().cpu().numpy(), W.I.float().cpu().numpy()
    pop = W.pop_last.cpu().numpy() if head else None
    return c_oracle.score_topk(Uw, Iw, np.arange(len(sub), dtype=np.int32), 50, head, pop, bip, bix, order=1, want_scores=True)


def assert_lists_match_oracle(keys, ridx, rval, sc, head):
    """Merged packed keys of the kernels against the oracle's lists: raw head bit-exact; popularity head 1e-5 on the values and
    any list disagreement a near-tie inside that tolerance (hardware v_exp_f32 vs libm expf in the last ulp)."""
    from pda_amd import ops
    idx, val = ops.unpack_keys(keys[:len(ridx)])
    if head == 0:
        np.testing.assert_array_equal(val, rval)
        np.testing.assert_array_equal(idx, ridx)
        return
    np.testing.assert_allclose(val, rval, rtol=TOL, atol=TOL)
    for r, k in np.argwhere(idx != ridx):
        a, b = idx[r, k], ridx[r, k]
        assert abs(sc[r, a] - sc[r, b]) <= TOL * max(1.0, abs(sc[r, b])), (r, k, a, b)



def run(ops, W, hist, users, head, prune, expect, **kw):
    """score_topk_keys + merge, with the identity of the sweep kernel that ran checked against `expect` (a dict of
    ops.kernel_identity fields)."""
    st = {}
    keys = ops.score_topk_keys(W.U, W.I, users, 50, head, W.pop_last if head else None, hist, prune=prune, stats=st, **kw)
    out = ops.topk_merge(keys, want="keys")
    ident = ops.kernel_identity(st["kernel_id"][0]) if "kernel_id" in st else {"generation": None}
    assert "error" not in st or int(st["error"][0]) == 0, ("the sweep reported a protocol error", int(st["error"][0]))
    for k, v in expect.items():
        assert ident.get(k) == v, (expect, ident)
    return out, st


def test_full_size_c3(dev, monkeypatch):
    """BASELINE config 3 at full size (1M users x 200k items, d = 128, real history CSR of 49M entries)."""
    from pda_amd import ops, synthetic
    W = synthetic.make_workload("c3", dev)
    hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
    POP, RAW = ops.HEAD_POP, ops.HEAD_RAW
    # ---- 16 384 users: the three sweep modes return identical keys; the exact fp32-MFMA kernel agrees on a 2 048-user subset
    users = torch.arange(500_000, 500_000 + 16384, dtype=torch.int32, device=dev)
    out = {}
    for name, prune, exp in (("natural", False, {"generation": 4, "geometry": "many"}), ("order", "order", {"generation": 4, "geometry": "huge", "early_stop": False}),
                             ("stop", True, {"generation": 4, "early_stop": True})):
        out[name], st = run(ops, W, hist, users, POP, prune, exp)
        if name == "stop":
            assert float(st["tiles_scored"][0]) < 0.2 * st["tiles_dense"]
    assert torch.equal(out["natural"], out["order"]) and torch.equal(out["natural"], out["stop"])
    sub = users[:2048].contiguous()
    exact = ops.topk_merge(ops.score_topk_keys(W.U, W.I, sub, 50, POP, W.pop_last, hist, impl="v1"), want="keys")
    assert torch.equal(exact, out["natural"][:2048])
    idx, val = ops.unpack_keys(out["stop"][:64])
    assert (np.diff(val, axis=1) <= 0).all() and (idx >= 0).all() and (idx < W.n_items).all()

    # ---- 131 072 users (the warm-up's masks come from their own kernel from 98 304 users on): generation 3 == generation 4
    big = torch.arange(200_000, 200_000 + 131072, dtype=torch.int32, device=dev)
    got = {}
    for kern, gen in (("v3", 3), ("v4", 4)):
        monkeypatch.setenv("PDA_SCORE_KERNEL", kern)
        monkeypatch.setenv("PDA_SCORE_LISTS", "lds")
        for prune in (True, "order"):
            got[(kern, prune)], _ = run(ops, W, hist, big, POP, prune, {"generation": gen, "d": 128, "head": 1})
    monkeypatch.delenv("PDA_SCORE_KERNEL")
    monkeypatch.delenv("PDA_SCORE_LISTS")
    ref = got[("v3", True)]
    for k, v in got.items():
        assert torch.equal(ref, v), k

    # ---- the operating point of bench.py's headline: a 262 144-user block, NO PDA_* variable set -- the library's own choice of
    # kernel and geometry, asserted: the dense sweep of the popularity head in visiting order runs the huge geometry
    # (sweep5_kernel<128, false, true, 256>), the natural-order sweep of that head the many-candidates geometry (<.., 3>), the raw head the funnel.
    # (a) the first 256 lists equal the oracle's on config 3's real history, both heads; (b) the users shared with the
    # 131 072-user block carry the same keys.
    huge = torch.arange(200_000, 200_000 + 262144, dtype=torch.int32, device=dev)
    oracle = {h: oracle_sample_lists(W, huge, h) for h in (0, 1)}
    k262 = {}
    for prune, exp in (("order", {"generation": 4, "geometry": "huge", "early_stop": False, "head": 1, "d": 128, "bf16": False}),
                       (True, {"generation": 4, "geometry": "lds", "early_stop": True, "head": 1}),
                       (False, {"generation": 4, "geometry": "many", "early_stop": False, "head": 1})):
        k262[prune], st = run(ops, W, hist, huge, POP, prune, exp)
        assert torch.equal(k262[prune][:131072], ref), prune
        assert_lists_match_oracle(k262[prune], *oracle[1], head=1)
        if prune == "order":
            assert int(st["tiles_scored"][0]) >= st["tiles_dense"]          # the dense sweep scored every tile
    # the raw head: the funnel (round 5: sweep7_kernel + expand7 / threshold7 / resolve7, pda_v7_funnel.h) from 32 768 users on, whatever the
    # sweep mode asked for (it visits the catalogue in its own random order); generation 4's many-candidates geometry when switched off.
    # Bit-exact scores and lists against the oracle, and against each other; next to no row may have needed the in-call exact fallback.
    raw = {}
    for prune, exp in ((None, {"generation": 4, "geometry": "funnel", "head": 0, "d": 128, "bf16": False}), (False, {"generation": 4, "geometry": "funnel", "head": 0})):
        raw[prune], st = run(ops, W, hist, huge, RAW, prune, exp)
        assert_lists_match_oracle(raw[prune], *oracle[0], head=0)
        assert int(st["fallback_rows"][0]) <= 262144 // 1000, int(st["fallback_rows"][0])
        assert 50 <= float(st["pairs_rescored"][0]) / 262144 <= 100          # exact rescorings per user: K + the pairs inside the bound's band
    assert torch.equal(raw[None], raw[False])
    monkeypatch.setenv("PDA_SCORE_FUNNEL", "0")
    raw_g4, _ = run(ops, W, hist, huge, RAW, None, {"generation": 4, "geometry": "many", "head": 0, "early_stop": False})
    monkeypatch.delenv("PDA_SCORE_FUNNEL")
    assert torch.equal(raw_g4, raw[None])
    # 98 304 users (the regrouped early-terminating sweep starts here; the dense sweep fills the chip with EIGHT item splits of the huge
    # geometry -- three rounds of 256 workgroups behind one shared warm-up: ops.huge_splits), again unforced; 53 248 users: nine splits (two
    # rounds); 16 384 users: sixteen; a 2 048-user block keeps the 256-user geometry
    mid = huge[:98304].contiguous()
    assert ops.huge_splits(98304, W.n_items) == 8 and ops.huge_splits(53248, W.n_items) == 9 and ops.huge_splits(16384, W.n_items) == 16
    assert ops.huge_splits(2048, W.n_items) == 0 and ops.huge_splits(262144, W.n_items) == 1
    for prune, exp in (("order", {"generation": 4, "geometry": "huge"}), (True, {"generation": 4, "early_stop": True})):
        k98, _ = run(ops, W, hist, mid, POP, prune, exp)
        assert torch.equal(k98, k262["order"][:98304]), prune
    k53, _ = run(ops, W, hist, huge[:53248].contiguous(), POP, "order", {"generation": 4, "geometry": "huge"})
    assert torch.equal(k53, k262["order"][:53248])
    k98r, _ = run(ops, W, hist, mid, RAW, None, {"generation": 4, "geometry": "funnel", "head": 0})       # (eight item splits)
    assert torch.equal(k98r, raw[None][:98304])
    k20r, _ = run(ops, W, hist, huge[:20000].contiguous(), RAW, None, {"generation": 4, "geometry": "funnel", "head": 0})
    assert torch.equal(k20r, raw[None][:20000])
    k2r, _ = run(ops, W, hist, huge[:2048].contiguous(), RAW, None, {"generation": 4, "geometry": "funnel", "head": 0})        # (the reference's own block size: 32 item splits)
    assert torch.equal(k2r, raw[None][:2048])
    k1r, _ = run(ops, W, hist, huge[:1000].contiguous(), RAW, None, {"generation": 4, "geometry": "funnel", "head": 0})        # (a partly filled 1 024-user tile: the funnel too since round 6)
    k64r, _ = run(ops, W, hist, huge[:64].contiguous(), RAW, None, {"generation": 4, "geometry": "funnel", "head": 0})
    assert torch.equal(k64r, raw[None][:64])
    assert torch.equal(k1r, raw[None][:1000])

    # ---- every other geometry forced on the same 262 144-user block: identical keys
    monkeypatch.setenv("PDA_SCORE_KERNEL", "v4")
    for geo, cases in (("lds", ((POP, "order"), (POP, True))), ("many", ((POP, "order"), (RAW, "order"))), ("huge", ((POP, "order"),))):
        monkeypatch.setenv("PDA_SCORE_LISTS", geo)
        for head, prune in cases:
            kg, _ = run(ops, W, hist, huge, head, prune, {"generation": 4, "head": head, "geometry": geo})
            assert torch.equal(kg, k262["order"] if head == POP else raw[None]), (geo, head, prune)
    torch.cuda.synchronize()


def test_full_size_c5_shard_bf16(dev, monkeypatch):
    """One rank's share of BASELINE config 5 at full size (250 000 item rows x d = 256, bf16 tables, the WHOLE 10M-row user
    table and its 600M-entry history CSR since round 6): the three sweep modes of both kernel generations return identical merged keys for
    8 192 users; a 256-user sample equals the oracle on the widened tables (popularity head: 1e-5 + near-tie rule); lists
    are sorted, in range and free of train items; then the 262 144-user block bench.py --workload c5shard times, unforced."""
    from pda_amd import ops, synthetic
    W = synthetic.make_workload("c5shard", dev, table_dtype=torch.bfloat16)
    assert W.U.dtype == torch.bfloat16 and W.I.shape == (250_000, 256)
    hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
    POP = ops.HEAD_POP
    users = torch.arange(300_000, 300_000 + 8192, dtype=torch.int32, device=dev)
    out = {}
    for kern, gen in (("v3", 3), ("v4", 4)):
        monkeypatch.setenv("PDA_SCORE_KERNEL", kern)
        for name, prune in (("natural", False), ("order", "order"), ("stop", True)):
            exp = {"generation": gen, "d": 256, "bf16": True} if not (kern == "v3" and prune is False) else {}       # (generation 3's natural-order path reports no stats)
            out[kern + name], st = run(ops, W, hist, users, POP, prune, exp)
            if name == "stop":
                assert float(st["tiles_scored"][0]) < st["tiles_dense"]          # (8 192 users are few user tiles: many item splits, each stopping on its own)
    monkeypatch.delenv("PDA_SCORE_KERNEL")
    ref = out["v3natural"]
    for k, v in out.items():
        assert torch.equal(ref, v), k
    idx, val = ops.unpack_keys(ref)
    assert (np.diff(val, axis=1) <= 0).all() and (idx >= 0).all() and (idx < W.n_items).all()
    for r in range(0, 8192, 512):
        u = 300_000 + r
        assert not set(idx[r]) & set(W.hist_indices[int(W.hist_indptr[u]):int(W.hist_indptr[u + 1])].cpu().numpy())
    ridx, rval, sc = oracle_sample_lists(W, users, 1)
    assert_lists_match_oracle(ref, ridx, rval, sc, head=1)
    # ---- the block bench.py --workload c5shard times (262 144 users), no PDA_* variable set: generation 4, d = 256, bf16 tables
    huge = torch.arange(300_000, 300_000 + 262144, dtype=torch.int32, device=dev)
    for prune, es in (("order", False), (True, True)):
        exp = {"generation": 4, "d": 256, "bf16": True, "early_stop": es, "head": 1}
        if not es:
            exp["geometry"] = "huge"              # the dense sweep of config 5's tables: sweep5_kernel<256, true, true, 128>
        k262, st = run(ops, W, hist, huge, POP, prune, exp)
        assert torch.equal(k262[:8192], ref), prune
        assert_lists_match_oracle(k262, ridx, rval, sc, head=1)
        if not es:
            assert int(st["tiles_scored"][0]) >= st["tiles_dense"]
    # ---- the raw head (rec_type main_branch) of the same block: the funnel at d = 256 (sweep7_kernel<256, true, 128>), unforced; the 8 192-user
    # block through generation 3 in natural order gives the same keys, and a 256-user sample equals the oracle
    RAW = ops.HEAD_RAW
    monkeypatch.setenv("PDA_SCORE_KERNEL", "v3")
    r8k, _ = run(ops, W, hist, users, RAW, False, {})
    monkeypatch.delenv("PDA_SCORE_KERNEL")
    st = {}
    r262 = ops.topk_merge(ops.score_topk_keys(W.U, W.I, huge, 50, RAW, None, hist, stats=st), want="keys")
    ident = ops.kernel_identity(st["kernel_id"][0])
    assert ident["generation"] == 4 and ident["geometry"] == "funnel" and ident["d"] == 256 and ident["bf16"] and int(st["error"][0]) == 0, ident
    assert int(st["fallback_rows"][0]) <= 262
    assert torch.equal(r262[:8192], r8k)
    ridx0, rval0, sc0 = oracle_sample_lists(W, users, 0)
    assert_lists_match_oracle(r262, ridx0, rval0, sc0, head=0)
    # ---- round 6: the WHOLE user table of config 5 (10 M x 256 bf16 = 5.1 GB, a 600 M-entry history CSR with int64 row pointers).  A block from
    # the top of the id range: every user-row gather (uid * 512 bytes) and every CSR read (indptr[uid] * 4 bytes) of these launches lies beyond 2^31
    # bytes.  Both sweep modes of the popularity head and the funnel agree with the oracle on a sample, with generation 3 on 8 192 users, and the
    # early-terminating keys equal the dense ones on all 262 144 rows.
    assert W.n_users == 10_000_000 and W.hist_indices.numel() > 560_000_000 and W.hist_indptr.dtype == torch.int64
    lo = 9_600_000
    assert lo > (1 << 23) and lo * 256 * 2 > (1 << 31) and int(W.hist_indptr[lo]) * 4 > (1 << 31)
    top = torch.arange(lo, lo + 262144, dtype=torch.int32, device=dev)
    tidx, tval, tsc = oracle_sample_lists(W, top, 1)
    tdense, _ = run(ops, W, hist, top, POP, "order", {"generation": 4, "d": 256, "bf16": True, "geometry": "huge", "head": 1})
    tstop, _ = run(ops, W, hist, top, POP, True, {"generation": 4, "d": 256, "bf16": True, "early_stop": True, "head": 1})
    assert torch.equal(tdense, tstop)
    assert_lists_match_oracle(tdense, tidx, tval, tsc, head=1)
    st = {}
    traw = ops.topk_merge(ops.score_topk_keys(W.U, W.I, top, 50, RAW, None, hist, stats=st), want="keys")
    ident = ops.kernel_identity(st["kernel_id"][0])
    assert ident["geometry"] == "funnel" and ident["d"] == 256 and int(st["error"][0]) == 0, ident
    tidx0, tval0, tsc0 = oracle_sample_lists(W, top, 0)
    assert_lists_match_oracle(traw, tidx0, tval0, tsc0, head=0)
    monkeypatch.setenv("PDA_SCORE_KERNEL", "v3")
    t8k, _ = run(ops, W, hist, top[:8192].contiguous(), RAW, False, {})
    monkeypatch.delenv("PDA_SCORE_KERNEL")
    assert torch.equal(traw[:8192], t8k)
    tidxs, _ = ops.unpack_keys(tdense[::4096])
    ipt = W.hist_indptr[top[::4096].long()].cpu().numpy(), W.hist_indptr[top[::4096].long() + 1].cpu().numpy()
    for r in range(tidxs.shape[0]):                               # train items never appear
        row = W.hist_indices[int(ipt[0][r]):int(ipt[1][r])].cpu().numpy()
        assert not set(tidxs[r]) & set(row)
    # ... and one SGD step on rows up there: the bf16 step kernel against torch on the gathered rows (fp32 masters take the update)
    g = torch.Generator(device=dev); g.manual_seed(3)
    B = 2048
    bu = (lo + torch.randperm(262144, generator=g, device=dev)[:B]).to(torch.int32)
    bp = torch.randint(0, W.n_items, (B,), generator=g, device=dev, dtype=torch.int32)
    bn = torch.randint(0, W.n_items, (B,), generator=g, device=dev, dtype=torch.int32)
    gU, gP, gN = (torch.zeros(B, 256, device=dev) for _ in range(3))
    loss = torch.zeros(3, device=dev)
    ops.bpr_step_bf16(W.U, W.I, bu, bp, bn, None, None, regs=1e-2, reg_div=B, mode=ops.UPD_NONE, grads_out=(gU, gP, gN), loss_acc=loss)
    ue, pe, ne = W.U[bu.long()].double(), W.I[bp.long()].double(), W.I[bn.long()].double()
    x = (ue * pe).sum(1) - (ue * ne).sum(1)
    sg = torch.sigmoid(x)
    gg = (-(1.0 / B) * sg * (1 - sg) / (sg + 1e-10))[:, None]
    torch.testing.assert_close(gU.double(), gg * (pe - ne) + 1e-2 / B * ue, atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(gP.double(), gg * ue + 1e-2 / B * pe, atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(float(loss[1]), float(-(torch.log(sg + 1e-10)).mean()), atol=1e-5, rtol=1e-5)



def test_a_c_caller_reaches_the_headline_kernel_through_the_plan(dev):
    """ctypes only -- no pda_amd.ops: what INTEGRATION.md section 2 tells a maintainer of the reference to write.  pda_score_topk_plan decides (item splits,
    geometry hint, visiting order, workspace); the prep, the call and the merge follow it; at 262 144 users of config 3 the identity word says
    sweep5_kernel (the huge geometry) ran, and the lists equal the ones pda_amd.ops returns."""
    import ctypes as C
    from pda_amd import _lib, ops, synthetic
    lib = _lib.load()
    W = synthetic.make_workload("c3", dev)
    nu, nI, d, K = 262144, W.n_items, W.d, 50
    users = torch.arange(200_000, 200_000 + nu, dtype=torch.int32, device=dev)
    p = _lib.ScorePlan()
    assert lib.pda_score_topk_plan(nu, nI, d, K, _lib.HEAD_POP, 2, 0, _lib.HIST_BY_USER_ID, C.byref(p)) == 0          # 2 = PDA_SWEEP_MODE_VISITING_ORDER: the dense headline sweep
    assert (p.path, p.n_splits, p.early_stop, p.order, p.prep_with_pop) == (4, 1, 128, 1, 1)
    order = torch.argsort(W.pop_last.abs(), descending=True, stable=True).to(torch.int32)                                # PDA_ORDER_BY_POPULARITY
    prep = torch.empty(lib.pda_item_prep4_bytes(nI, d), dtype=torch.uint8, device=dev)
    ptr, sp = _lib.ptr, _lib.stream_ptr
    assert lib.pda_item_prep4_f32(ptr(W.I), ptr(W.pop_last), ptr(order), nI, d, ptr(prep), sp()) == 0
    ws = torch.empty(p.workspace_bytes, dtype=torch.uint8, device=dev)
    keys = torch.empty((p.keys_rows, K), dtype=torch.int64, device=dev)
    assert lib.pda_score_topk4_f32(ptr(W.U), ptr(W.I), ptr(prep), ptr(W.pop_last), ptr(users), nu, 0, nI, d, ptr(W.hist_indptr), ptr(W.hist_indices),
                                   _lib.HIST_BY_USER_ID, K, _lib.HEAD_POP, p.early_stop, p.n_splits, ptr(keys), ptr(ws), sp()) == 0
    torch.cuda.synchronize()
    ident = ops.kernel_identity(ws[16:20].view(torch.int32)[0])
    assert ident["generation"] == 4 and ident["geometry"] == "huge" and ident["d"] == 128 and int(ws[0:4].view(torch.int32)[0]) == 0, ident
    hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
    ref = ops.topk_merge(ops.score_topk_keys(W.U, W.I, users, K, ops.HEAD_POP, W.pop_last, hist, prune="order"), want="keys")
    assert torch.equal(keys.view(nu, K), ref)
