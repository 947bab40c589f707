"""GPU parity: fused BPR triplet step (pda_bpr_step_f32), Adam sweeps, metrics and the sampler vs the oracle.
Tolerance 1e-5 absolute on loss / gradients / updated rows (north_star), float64 oracle as the truth."""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import pda_oracle as po

pytestmark = pytest.mark.gpu
TOL = 1e-5


def triplets(rng, nU, nI, B, dup_items=True):
    users = rng.permutation(nU)[:B].astype(np.int32)          # unique per batch (rd.sample)
    hi = max(2, nI // 20) if dup_items else nI                 # force repeated item rows inside the batch
    pos = rng.integers(0, hi, B).astype(np.int32)
    neg = rng.integers(0, hi, B).astype(np.int32)
    return users, pos, neg


def to(dev, *xs):
    return [None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in xs]


@pytest.mark.parametrize("d", [32, 64, 128, 256])
@pytest.mark.parametrize("with_pop", [False, True])
def test_loss_and_gradients(dev, d, with_pop):
    from pda_amd import ops
    rng = np.random.default_rng(d + int(with_pop))
    nU, nI, B, regs = 3000, 900, 2048, 1e-2
    U = (rng.standard_normal((nU, d)) * 0.3).astype(np.float32)   # large enough that ELU sees both branches
    I = (rng.standard_normal((nI, d)) * 0.3).astype(np.float32)
    users, pos, neg = triplets(rng, nU, nI, B)
    pp = pn = None
    if with_pop:
        pp = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
        pn = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
    fw = po.bpr_forward(U, I, users, pos, neg, pp, pn)
    ref_loss = po.bpr_loss(fw, regs, B)
    rdu, rdp, rdn = po.bpr_grads(fw, regs, B, pp, pn)

    Ut, It, ut, pt, nt, ppt, pnt = to(dev, U, I, users, pos, neg, pp, pn)
    gu, gp, gn = (torch.empty(B, d, device=dev) for _ in range(3))
    loss = torch.zeros(3, device=dev)
    ops.bpr_step(Ut, It, ut, pt, nt, ppt, pnt, regs=regs, reg_div=B, mode=ops.UPD_NONE, grads_out=(gu, gp, gn),
                 loss_acc=loss)
    np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, atol=TOL, rtol=TOL)
    for g, r in ((gu, rdu), (gp, rdp), (gn, rdn)):
        np.testing.assert_allclose(g.cpu().numpy(), r, atol=TOL)
    assert torch.equal(Ut.cpu(), torch.from_numpy(U)) and torch.equal(It.cpu(), torch.from_numpy(I))  # no update


@pytest.mark.parametrize("with_pop", [False, True])
def test_sgd_fused_update_with_duplicate_items(dev, with_pop):
    from pda_amd import ops
    rng = np.random.default_rng(17)
    nU, nI, d, B, regs, lr = 5000, 400, 64, 2048, 1e-2, 0.05
    U = (rng.standard_normal((nU, d)) * 0.2).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * 0.2).astype(np.float32)
    users, pos, neg = triplets(rng, nU, nI, B)
    pp = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32) if with_pop else None
    pn = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32) if with_pop else None
    U1, I1, _, ref_loss = po.train_step(U, I, users, pos, neg, pp, pn, regs, B, lr, optimizer="sgd")
    Ut, It, ut, pt, nt, ppt, pnt = to(dev, U, I, users, pos, neg, pp, pn)
    loss = torch.zeros(3, device=dev)
    ops.bpr_step(Ut, It, ut, pt, nt, ppt, pnt, regs=regs, reg_div=B, lr=lr, mode=ops.UPD_SGD_FUSED, loss_acc=loss)
    np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, atol=TOL, rtol=TOL)
    np.testing.assert_allclose(Ut.cpu().numpy(), U1, atol=TOL)
    np.testing.assert_allclose(It.cpu().numpy(), I1, atol=TOL)


@pytest.mark.parametrize("d", [32, 64, 128, 256])
def test_sgd_fused_plain_user_store_with_distinct_users(dev, d):
    """PDA_UPD_USERS_DISTINCT: with the reference sampler's contract (users drawn without replacement, MF/train_new_api.py:380-381)
    the user rows take plain stores.  The step must still equal the oracle's SGD step, and U the atomic variant's to rounding (not
    bit for bit: inside one launch a triplet may read an item row another triplet's atomics are half way through -- hogwild)."""
    from pda_amd import ops
    rng = np.random.default_rng(23 + d)
    nU, nI, B, regs, lr = 5000, 400, 2048, 1e-2, 0.05
    U = (rng.standard_normal((nU, d)) * 0.2).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * 0.2).astype(np.float32)
    users = rng.permutation(nU)[:B].astype(np.int32)
    pos = rng.integers(0, nI, B).astype(np.int32)
    neg = rng.integers(0, nI, B).astype(np.int32)
    pp = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
    pn = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
    U1, I1, _, ref_loss = po.train_step(U, I, users, pos, neg, pp, pn, regs, B, lr, optimizer="sgd")
    res = []
    for flag in (True, False):
        Ut, It, ut, pt, nt, ppt, pnt = to(dev, U, I, users, pos, neg, pp, pn)
        loss = torch.zeros(3, device=dev)
        ops.bpr_step(Ut, It, ut, pt, nt, ppt, pnt, regs=regs, reg_div=B, lr=lr, mode=ops.UPD_SGD_FUSED, loss_acc=loss, users_distinct=flag)
        np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, atol=TOL, rtol=TOL)
        np.testing.assert_allclose(Ut.cpu().numpy(), U1, atol=TOL)
        np.testing.assert_allclose(It.cpu().numpy(), I1, atol=TOL)
        res.append(Ut.cpu())
    np.testing.assert_allclose(res[0].numpy(), res[1].numpy(), atol=1e-6)


def _hot_batch(rng, nU, nI, B, hot_share=0.3):
    """A batch in which one positive item carries `hot_share` of the triplets and users repeat (B > nU)."""
    users = rng.integers(0, nU, B).astype(np.int32)
    pos = rng.integers(0, nI, B).astype(np.int32)
    pos[rng.random(B) < hot_share] = 7
    neg = rng.integers(0, nI, B).astype(np.int32)
    return users, pos, neg


@pytest.mark.parametrize("lr", [0.05, 1.0, 5.0])
def test_exact_sgd_step_on_a_hot_item_batch_at_realistic_lr(dev, lr):
    """the fused in-kernel update is asynchronous inside a launch (hogwild).  The EXACT step -- gradients of the
    whole batch against the unchanged tables, then one scatter -- must equal the oracle's mini-batch SGD step also with a
    hot positive (30 % of the batch), repeated users and a large learning rate, where cross terms would show."""
    from pda_amd import ops
    rng = np.random.default_rng(31)
    nU, nI, d, B, regs = 600, 300, 64, 2048, 1e-2
    U = (rng.standard_normal((nU, d)) * 0.3).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * 0.3).astype(np.float32)
    users, pos, neg = _hot_batch(rng, nU, nI, B)
    pp = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
    pn = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
    U1, I1, _, ref_loss = po.train_step(U, I, users, pos, neg, pp, pn, regs, B, lr, optimizer="sgd")
    Ut, It, ut, pt, nt, ppt, pnt = to(dev, U, I, users, pos, neg, pp, pn)
    loss = torch.zeros(3, device=dev)
    ops.sgd_step_exact(Ut, It, ut, pt, nt, ppt, pnt, regs=regs, reg_div=B, lr=lr, loss_acc=loss)
    scale = max(1.0, lr)                                        # the hot row moves by O(lr): tolerance relative to that
    np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, atol=TOL, rtol=TOL)
    np.testing.assert_allclose(Ut.cpu().numpy(), U1, atol=TOL * scale)
    np.testing.assert_allclose(It.cpu().numpy(), I1, atol=TOL * scale)
    # and the fused one-launch update is a DIFFERENT (asynchronous) step: the same loss -- the forward pass of the batch is
    # taken before most updates land --, tables equal to the exact step up to O(lr^2) cross terms
    Uf, If = to(dev, U, I)
    lossf = torch.zeros(3, device=dev)
    ops.bpr_step(Uf, If, ut, pt, nt, ppt, pnt, regs=regs, reg_div=B, lr=lr, mode=ops.UPD_SGD_FUSED, loss_acc=lossf)
    dev_f = float((If.cpu() - torch.from_numpy(I1)).abs().max())
    assert dev_f <= 0.5 * lr * lr + 1e-4, dev_f                 # bounded by the cross terms, but NOT held to 1e-5


def adam_close(got, ref):
    """Tables after Adam steps against the float64 oracle: 1e-5 -- except where a summed gradient component is itself of the size
    of Adam's epsilon (1e-8): there lr m / (sqrt(v) + eps) turns the ORDER of the fp32 atomic adds of a repeated row into
    1e-5 .. 1e-4 of x (seen once in ~20 runs), so a handful of elements get that much."""
    err = np.abs(got - ref)
    assert (err > TOL).mean() < 1e-3 and err.max() < 2e-4, ((err > TOL).mean(), err.max())


def test_reference_faithful_adam_three_steps(dev):
    """TF-1.14 Adam: m,v decay and the variable update touch EVERY row each step [TF-ext]; three steps
    exercise that on rows that were touched once and then left alone."""
    from pda_amd import ops
    rng = np.random.default_rng(23)
    nU, nI, d, B, regs, lr = 2500, 700, 64, 1024, 1e-2, 1e-2
    U = (rng.standard_normal((nU, d)) * 0.1).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * 0.1).astype(np.float32)
    Ut, It = to(dev, U, I)
    st = {k: torch.zeros_like(t) for k, t in (("mU", Ut), ("vU", Ut), ("gU", Ut), ("mI", It), ("vI", It), ("gI", It))}
    Ur, Ir, state = U.astype(np.float64), I.astype(np.float64), None
    for t in (1, 2, 3):
        users, pos, neg = triplets(rng, nU, nI, B)
        pp = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
        pn = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
        Ur, Ir, state, ref_loss = po.train_step(Ur, Ir, users, pos, neg, pp, pn, regs, B, lr, "adam", state, t)
        loss = torch.zeros(3, device=dev)
        ops.bpr_step(Ut, It, *to(dev, users, pos, neg, pp, pn), regs=regs, reg_div=B, mode=ops.UPD_DENSE_GRAD,
                     gU=st["gU"], gI=st["gI"], loss_acc=loss)
        lr_t = ops.adam_lr_t(lr, t)
        ops.adam_dense_sweep(Ut, st["mU"], st["vU"], st["gU"], lr_t)
        ops.adam_dense_sweep(It, st["mI"], st["vI"], st["gI"], lr_t)
        np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, atol=TOL, rtol=TOL)
        assert float(st["gU"].abs().max()) == 0.0 and float(st["gI"].abs().max()) == 0.0   # accumulator reset
    for got, ref in ((Ut, Ur), (It, Ir)):
        adam_close(got.cpu().numpy(), ref)
    np.testing.assert_allclose(st["mI"].cpu().numpy(), state["mI"], atol=TOL)
    np.testing.assert_allclose(st["vI"].cpu().numpy(), state["vI"], atol=TOL)


@pytest.mark.parametrize("repeats", [False, True])
def test_exact_lazy_adam_equals_the_dense_sweep_bit_for_bit(dev, repeats):
    """pda_adam_lazy_f32 / pda_adam_lazy_sync_f32: the reference's dense-decay Adam WITHOUT the sweep.  Twelve steps on small
    batches (most rows idle most of the time, item rows repeated inside a batch, one batch replayed after a long idle gap);
    after the final sync every element of U, I, m, v equals the dense-sweep path bit for bit, and the oracle's dense-decay
    Adam (po.train_step) to 1e-5 -- including rows touched once and then left alone."""
    from pda_amd import ops
    rng = np.random.default_rng(41)
    nU, nI, d, B, regs, lr, N = 900, 400, 64, 96, 1e-2, 1e-2, 12
    assert 2 * B <= nI
    U = (rng.standard_normal((nU, d)) * 0.1).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * 0.1).astype(np.float32)
    Ud, Id = to(dev, U, I)                        # dense sweeps
    Ul, Il = to(dev, U, I)                        # lazy
    z = torch.zeros_like
    sd = {k: z(t) for k, t in (("mU", Ud), ("vU", Ud), ("gU", Ud), ("mI", Id), ("vI", Id), ("gI", Id))}
    sl = {k: z(t) for k, t in (("mU", Ul), ("vU", Ul), ("gU", Ul), ("mI", Il), ("vI", Il), ("gI", Il))}
    lz = ops.LazyAdamState(nU, nI, lr, dev)
    Ur, Ir, state = U.astype(np.float64), I.astype(np.float64), None
    first = None
    for t in range(1, N + 1):
        users = rng.permutation(nU)[:B].astype(np.int32)
        if repeats:
            pos = rng.integers(0, 40 if t % 3 else nI, B).astype(np.int32)      # a hot head of items: repeats inside the batch
            neg = rng.integers(0, nI, B).astype(np.int32)
        else:
            pi = rng.permutation(nI).astype(np.int32)                            # every item row at most once per batch
            pos, neg = pi[:B], pi[B:2 * B]
        if first is None:
            first = (users, pos, neg)
        if t == N:
            users, pos, neg = first                                              # rows idle since step 1 come back
        pp = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
        pn = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
        Ur, Ir, state, ref_loss = po.train_step(Ur, Ir, users, pos, neg, pp, pn, regs, B, lr, "adam", state, t)
        args = to(dev, users, pos, neg, pp, pn)
        ld, ll = torch.zeros(3, device=dev), torch.zeros(3, device=dev)
        ops.bpr_step(Ud, Id, *args, regs=regs, reg_div=B, mode=ops.UPD_DENSE_GRAD, gU=sd["gU"], gI=sd["gI"], loss_acc=ld)
        ops.adam_dense_sweep2(Ud, sd["mU"], sd["vU"], sd["gU"], Id, sd["mI"], sd["vI"], sd["gI"], ops.adam_lr_t(lr, t))
        ops.adam_lazy(0, lz, Ul, sl["mU"], sl["vU"], sl["gU"], Il, sl["mI"], sl["vI"], sl["gI"], *args[:3], t)
        ops.bpr_step(Ul, Il, *args, regs=regs, reg_div=B, mode=ops.UPD_DENSE_GRAD, gU=sl["gU"], gI=sl["gI"], loss_acc=ll)
        ops.adam_lazy(1, lz, Ul, sl["mU"], sl["vU"], sl["gU"], Il, sl["mI"], sl["vI"], sl["gI"], *args[:3], t)
        np.testing.assert_allclose(ll.cpu().numpy(), ref_loss, atol=TOL, rtol=TOL)
        assert float(sl["gU"].abs().max()) == 0.0 and float(sl["gI"].abs().max()) == 0.0
    idle = int((lz.lastU < N).sum()) + int((lz.lastI < N).sum())
    assert idle > 300                              # most rows have NOT been touched by the last step: the sync has work to do
    assert not torch.equal(Ul, Ud)
    ops.adam_lazy_sync(lz, Ul, sl["mU"], sl["vU"], Il, sl["mI"], sl["vI"], N)
    assert int(lz.lastU.min()) == N and int(lz.lastI.min()) == N
    # (the summed gradient of a repeated row is an atomic sum whose order differs between two launches: with repeats the two
    # runs agree to rounding only, without them bit for bit)
    for a, b, ref in ((Ul, Ud, Ur), (Il, Id, Ir), (sl["mU"], sd["mU"], state["mU"]), (sl["vU"], sd["vU"], state["vU"]),
                      (sl["mI"], sd["mI"], state["mI"]), (sl["vI"], sd["vI"], state["vI"])):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=2e-7, rtol=1e-5)
        # against the float64 oracle: 1e-5, except where a gradient component is itself of the size of Adam's epsilon
        # (1e-8: lr m / (sqrt(v) + eps) then turns the fp32 rounding of g into 1e-5 .. 1e-4 of x -- the dense sweep, which is
        # the arithmetic compared bit for bit above, differs from the oracle in exactly the same elements)
        err = np.abs(a.cpu().numpy() - ref)
        assert (err > TOL).mean() < 1e-3 and err.max() < 2e-4, ((err > TOL).mean(), err.max())
        if not repeats:
            assert torch.equal(a, b)
    Uc = Ul.clone()
    ops.adam_lazy_sync(lz, Ul, sl["mU"], sl["vU"], Il, sl["mI"], sl["vI"], N)      # idempotent
    assert torch.equal(Ul, Uc)


def test_exact_lazy_adam_long_idle_gap(dev):
    """3 000 idle steps replayed in one go (the update falls below half an ulp after ~150 steps; from there only m and v
    decay) == 3 000 dense sweeps with zero gradient, bit for bit -- also for rows whose x is tiny or zero (they keep moving
    for longer) and for zero moments."""
    from pda_amd import ops
    rng = np.random.default_rng(43)
    n, d, T, lr = 96, 64, 3000, 1e-2
    var = (rng.standard_normal((n, d)) * 0.1).astype(np.float32)
    m = (rng.standard_normal((n, d)) * 1e-3).astype(np.float32)
    v = (rng.uniform(0, 1, (n, d)) ** 4 * 1e-5).astype(np.float32)
    var[0] = 0.0
    var[1] *= 1e-20
    m[2] = 0.0
    v[3] = 0.0
    vd, md, vvd = to(dev, var, m, v)
    vl, ml, vvl = to(dev, var, m, v)
    g = torch.zeros_like(vd)
    for t in range(1, T + 1):
        ops.adam_dense_sweep(vd, md, vvd, g, ops.adam_lr_t(lr, t))
    lz = ops.LazyAdamState(n, n, lr, dev)
    dummy = [x.clone() for x in (vl, ml, vvl)]
    ops.adam_lazy_sync(lz, vl, ml, vvl, *dummy, T)
    assert torch.equal(vl, vd) and torch.equal(ml, md) and torch.equal(vvl, vvd)
    assert not torch.equal(vl, torch.from_numpy(var).to(dev))


@pytest.mark.parametrize("fast", [False, True])
def test_lazy_adam_idle_gaps_inside_the_first_hundred_steps(dev, fast):
    """The bias-corrected rate lr_k = lr sqrt(1 - b2^k) / (1 - b1^k) is NOT monotone over the first steps (it falls until k ~ 10,
    then rises by up to 1.2 % per step): rows that go idle at step a and come back at step b, for many (a, b) inside 1 .. 100,
    replayed == swept (exact replay: bit for bit; fast replay: x to 1e-6, m and v to 1e-4 relative)."""
    from pda_amd import ops
    rng = np.random.default_rng(47)
    n, d, lr, T = 128, 64, 1e-2, 100
    var = (rng.standard_normal((n, d)) * 0.1).astype(np.float32)
    v0 = (rng.uniform(0, 1, (n, d)) ** 4 * 1e-5).astype(np.float32)
    # moments as Adam produces them: |m| <= ~3 sqrt(v) (m is a (1 - b1)-average of gradients whose squares make up v); the exact
    # replay is also run on wild pairs (m >> sqrt(v): x moves by hundreds) in test_exact_lazy_adam_long_idle_gap
    m0 = ((np.sqrt(v0) * rng.uniform(-2, 2, (n, d))) if fast else rng.standard_normal((n, d)) * 1e-3).astype(np.float32)
    starts = rng.integers(0, 60, n)                 # row r is current for step starts[r] and idle until ends[r]
    ends = np.minimum(T, starts + rng.integers(1, 60, n))
    vd, md, vvd = to(dev, var, m0, v0)
    g = torch.zeros_like(vd)
    # dense: sweep step k touches only the rows with starts < k <= ends (emulated by restoring the others)
    for k in range(1, T + 1):
        act = torch.from_numpy(((starts < k) & (k <= ends))).to(dev)
        keep = (vd.clone(), md.clone(), vvd.clone())
        ops.adam_dense_sweep(vd, md, vvd, g, ops.adam_lr_t(lr, k))
        for t_, k_ in zip((vd, md, vvd), keep):
            t_[~act] = k_[~act]
    vl, ml, vvl = to(dev, var, m0, v0)
    lz = ops.LazyAdamState(n, n, lr, dev, fast=fast)
    lz.lastU.copy_(torch.from_numpy(starts.astype(np.int32)))
    lz.lastI.fill_(T)
    dummy = [x.clone() for x in (vl, ml, vvl)]
    # bring every row to ITS end step: rows grouped by end step, synced with t = end (others are pushed back afterwards)
    for e in np.unique(ends):
        rows = torch.from_numpy(np.nonzero(ends == e)[0]).to(dev)
        sub = [t_[rows].contiguous() for t_ in (vl, ml, vvl)]
        st = ops.LazyAdamState(len(rows), len(rows), lr, dev, fast=fast)
        st.lastU.copy_(lz.lastU[rows])
        st.lastI.fill_(int(e))
        d2 = [x.clone() for x in sub]
        ops.adam_lazy_sync(st, sub[0], sub[1], sub[2], *d2, int(e))
        for t_, s_ in zip((vl, ml, vvl), sub):
            t_[rows] = s_
    if fast:
        np.testing.assert_allclose(vl.cpu().numpy(), vd.cpu().numpy(), atol=1e-6, rtol=1e-6)
        np.testing.assert_allclose(ml.cpu().numpy(), md.cpu().numpy(), rtol=1e-4, atol=1e-12)
        np.testing.assert_allclose(vvl.cpu().numpy(), vvd.cpu().numpy(), rtol=1e-4, atol=1e-14)
    else:
        assert torch.equal(vl, vd) and torch.equal(ml, md) and torch.equal(vvl, vvd)


def test_fast_lazy_adam_long_idle_gap_and_training(dev):
    """PDA_ADAM_REPLAY_FAST: (a) 3 000 idle steps in one go against 3 000 dense sweeps: x within 1e-6, m and v within 1e-4
    relative -- zero / tiny rows, zero moments and moments at the size of Adam's epsilon included; (b) twelve training steps with
    repeats against the dense-sweep path: every tensor within the same bounds."""
    from pda_amd import ops
    rng = np.random.default_rng(53)
    n, d, T, lr = 96, 64, 3000, 1e-2
    var = (rng.standard_normal((n, d)) * 0.1).astype(np.float32)
    v = (rng.uniform(0, 1, (n, d)) ** 4 * 1e-5).astype(np.float32)
    m = (np.sqrt(v) * rng.uniform(-2, 2, (n, d))).astype(np.float32)        # |m| <= ~3 sqrt(v), as Adam's moments are
    var[0] = 0.0
    var[1] *= 1e-20
    m[2] = 0.0
    v[3] = 0.0
    m[3] = 0.0
    m[4] = (rng.standard_normal(d) * 1e-9).astype(np.float32)               # gradients of the size of epsilon: sqrt(v) ~ eps
    v[4] = (m[4] * 3) ** 2
    m[5] = (rng.standard_normal(d) * 1e-3).astype(np.float32)               # a wild pair: m >> sqrt(v), x moves by tens
    v[5] = 1e-12
    vd, md, vvd = to(dev, var, m, v)
    vl, ml, vvl = to(dev, var, m, v)
    g = torch.zeros_like(vd)
    for t in range(1, T + 1):
        ops.adam_dense_sweep(vd, md, vvd, g, ops.adam_lr_t(lr, t))
    lz = ops.LazyAdamState(n, n, lr, dev, fast=True)
    dummy = [x.clone() for x in (vl, ml, vvl)]
    ops.adam_lazy_sync(lz, vl, ml, vvl, *dummy, T)
    np.testing.assert_allclose(vl.cpu().numpy(), vd.cpu().numpy(), atol=1e-6, rtol=1e-6)
    np.testing.assert_allclose(ml.cpu().numpy(), md.cpu().numpy(), rtol=1e-4, atol=1e-30)
    np.testing.assert_allclose(vvl.cpu().numpy(), vvd.cpu().numpy(), rtol=1e-4, atol=1e-30)
    assert not torch.equal(vl, torch.from_numpy(var).to(dev))
    # (b) training
    nU, nI, B, regs, N = 900, 400, 96, 1e-2, 12
    U = (rng.standard_normal((nU, d)) * 0.1).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * 0.1).astype(np.float32)
    Ud, Id = to(dev, U, I)
    Ul, Il = to(dev, U, I)
    z = torch.zeros_like
    sd = {k: z(t) for k, t in (("mU", Ud), ("vU", Ud), ("gU", Ud), ("mI", Id), ("vI", Id), ("gI", Id))}
    sl = {k: z(t) for k, t in (("mU", Ul), ("vU", Ul), ("gU", Ul), ("mI", Il), ("vI", Il), ("gI", Il))}
    lz = ops.LazyAdamState(nU, nI, lr, dev, fast=True)
    for t in range(1, N + 1):
        users = rng.permutation(nU)[:B].astype(np.int32)
        pos = rng.integers(0, 40 if t % 3 else nI, B).astype(np.int32)
        neg = rng.integers(0, nI, B).astype(np.int32)
        pp = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
        pn = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
        args = to(dev, users, pos, neg, pp, pn)
        ld, ll = torch.zeros(3, device=dev), torch.zeros(3, device=dev)
        ops.bpr_step(Ud, Id, *args, regs=regs, reg_div=B, mode=ops.UPD_DENSE_GRAD, gU=sd["gU"], gI=sd["gI"], loss_acc=ld)
        ops.adam_dense_sweep2(Ud, sd["mU"], sd["vU"], sd["gU"], Id, sd["mI"], sd["vI"], sd["gI"], ops.adam_lr_t(lr, t))
        ops.adam_lazy(0, lz, Ul, sl["mU"], sl["vU"], sl["gU"], Il, sl["mI"], sl["vI"], sl["gI"], *args[:3], t)
        ops.bpr_step(Ul, Il, *args, regs=regs, reg_div=B, mode=ops.UPD_DENSE_GRAD, gU=sl["gU"], gI=sl["gI"], loss_acc=ll)
        ops.adam_lazy(1, lz, Ul, sl["mU"], sl["vU"], sl["gU"], Il, sl["mI"], sl["vI"], sl["gI"], *args[:3], t)
        np.testing.assert_allclose(ll.cpu().numpy(), ld.cpu().numpy(), atol=TOL, rtol=TOL)
    ops.adam_lazy_sync(lz, Ul, sl["mU"], sl["vU"], Il, sl["mI"], sl["vI"], N)
    for a, b in ((Ul, Ud), (Il, Id)):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=2e-6, rtol=0)
    # (m = b1 m + (1 - b1) g can cancel to a few 1e-9 where both terms are 1e-5: absolute floors at 1e-6 of the typical size)
    for k in ("mU", "mI"):
        np.testing.assert_allclose(sl[k].cpu().numpy(), sd[k].cpu().numpy(), rtol=2e-4, atol=1e-10)
    for k in ("vU", "vI"):
        np.testing.assert_allclose(sl[k].cpu().numpy(), sd[k].cpu().numpy(), rtol=2e-4, atol=1e-16)


def test_lazy_adam_with_the_step_counter_on_the_device(dev):
    """pda_adam_lazy_dev_f32: sixteen steps with t read from device memory (two alternating counter slots), eager and replayed from a
    HIP graph, equal the same steps with t passed from the host (bit for bit: the same kernels, the same arithmetic)."""
    from pda_amd import ops
    rng = np.random.default_rng(61)
    nU, nI, d, B, regs, lr, N = 900, 400, 64, 96, 1e-2, 1e-2, 16
    U = (rng.standard_normal((nU, d)) * 0.1).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * 0.1).astype(np.float32)
    batches = []
    for t in range(N):
        pi = rng.permutation(nI).astype(np.int32)
        batches.append(to(dev, rng.permutation(nU)[:B].astype(np.int32), pi[:B], pi[B:2 * B], (rng.uniform(0, 1, B) ** 0.22).astype(np.float32),
                          (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)))
    z = torch.zeros_like

    def fresh():
        Ut, It = to(dev, U, I)
        return Ut, It, [z(Ut), z(Ut), z(Ut), z(It), z(It), z(It)], ops.LazyAdamState(nU, nI, lr, dev), torch.zeros(3, device=dev)
    Ua, Ia, sa, la, lossa = fresh()
    for t in range(1, N + 1):
        b = batches[t - 1]
        ops.adam_lazy(0, la, Ua, sa[0], sa[1], sa[2], Ia, sa[3], sa[4], sa[5], *b[:3], t)
        ops.bpr_step(Ua, Ia, *b, regs=regs, reg_div=B, mode=ops.UPD_DENSE_GRAD, gU=sa[2], gI=sa[5], loss_acc=lossa)
        ops.adam_lazy(1, la, Ua, sa[0], sa[1], sa[2], Ia, sa[3], sa[4], sa[5], *b[:3], t)
    for graph in (False, True):
        Ub, Ib, sb, lb, lossb = fresh()
        t_dev = torch.tensor([1, 0], dtype=torch.int32, device=dev)
        lb.rates(N + 8)

        def step(i):
            b = batches[i]
            ops.adam_lazy_dev(0, lb, Ub, sb[0], sb[1], sb[2], Ib, sb[3], sb[4], sb[5], *b[:3], t_dev, i & 1, N + 8)
            ops.bpr_step(Ub, Ib, *b, regs=regs, reg_div=B, mode=ops.UPD_DENSE_GRAD, gU=sb[2], gI=sb[5], loss_acc=lossb)
            ops.adam_lazy_dev(1, lb, Ub, sb[0], sb[1], sb[2], Ib, sb[3], sb[4], sb[5], *b[:3], t_dev, i & 1, N + 8)
        if not graph:
            for i in range(N):
                step(i)
        else:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(s):
                with torch.cuda.graph(g):
                    for i in range(N // 2):
                        step(i)                              # eight steps per replay: the second replay must go on at step 9
            torch.cuda.current_stream().wait_stream(s)
            # (the capture itself launched nothing; the second half replays the same graph on the batches of the first: feed both halves the same batches)
            g.replay()
            torch.cuda.synchronize()
            assert int(t_dev[0]) == N // 2 + 1
            continue_ok = True
            # compare after the first replay against the host-counter run truncated to eight steps
            Uc, Ic, sc_, lc, lossc = fresh()
            for t in range(1, N // 2 + 1):
                b = batches[t - 1]
                ops.adam_lazy(0, lc, Uc, sc_[0], sc_[1], sc_[2], Ic, sc_[3], sc_[4], sc_[5], *b[:3], t)
                ops.bpr_step(Uc, Ic, *b, regs=regs, reg_div=B, mode=ops.UPD_DENSE_GRAD, gU=sc_[2], gI=sc_[5], loss_acc=lossc)
                ops.adam_lazy(1, lc, Uc, sc_[0], sc_[1], sc_[2], Ic, sc_[3], sc_[4], sc_[5], *b[:3], t)
            assert torch.equal(Ub, Uc) and torch.equal(Ib, Ic) and torch.equal(lb.lastU, lc.lastU) and continue_ok
            g.replay()                                        # steps 9 .. 16 of the optimiser on the batches 1 .. 8 again
            torch.cuda.synchronize()
            assert int(t_dev[0]) == N + 1 and int(lb.lastU.max()) == N
            continue
        assert int(t_dev[0]) == N + 1
        assert torch.equal(Ub, Ua) and torch.equal(Ib, Ia)
        for x, y in zip(sb, sa):
            assert torch.equal(x, y)


def test_lazy_adam_rows_matches_dense_on_touched_rows(dev):
    from pda_amd import ops
    rng = np.random.default_rng(29)
    n, d = 1000, 64
    var = (rng.standard_normal((n, d))).astype(np.float32)
    g = np.zeros((n, d), np.float32)
    rows = np.sort(rng.permutation(n)[:200]).astype(np.int32)
    g[rows] = rng.standard_normal((200, d)).astype(np.float32)
    vt, gt, rt = to(dev, var, g, rows)
    m, v = torch.zeros_like(vt), torch.zeros_like(vt)
    ops.adam_rows(vt, m, v, gt, rt, ops.adam_lr_t(1e-2, 1))
    r_var, r_m, r_v = po.adam_dense_decay_step(var.astype(np.float64), 0.0, 0.0, g.astype(np.float64), 1, 1e-2)
    np.testing.assert_allclose(vt.cpu().numpy()[rows], r_var[rows], atol=TOL)
    untouched = np.setdiff1d(np.arange(n), rows)
    np.testing.assert_array_equal(vt.cpu().numpy()[untouched], var[untouched])
    assert float(gt.abs().max()) == 0.0


def test_metrics_kernel(dev):
    from pda_amd import ops
    rng = np.random.default_rng(31)
    n, K = 1000, 50
    topk = np.stack([rng.permutation(3000)[:K] for _ in range(n)]).astype(np.int32)
    targets = [rng.permutation(3000)[:rng.integers(1, 80)].astype(np.int32) for _ in range(n)]
    for r in range(0, n, 3):    # make hits likely
        targets[r][: min(5, len(targets[r]))] = topk[r][rng.permutation(K)[: min(5, len(targets[r]))]]
    indptr = np.zeros(n + 1, np.int64)
    indptr[1:] = np.cumsum([len(t) for t in targets])
    flat = np.concatenate(targets)
    Ks = np.array([20, 50], np.int32)
    sums = ops.metrics_sums(*to(dev, topk, indptr, flat, Ks)).cpu().numpy()
    ref = {k: np.zeros(2) for k in ("precision", "recall", "ndcg", "hit_ratio")}
    for r in range(n):
        one = po.get_performance(targets[r].tolist(), topk[r], Ks.tolist())
        for k in ref:
            ref[k] += one[k]
    for row, k in enumerate(("precision", "recall", "ndcg", "hit_ratio")):
        np.testing.assert_allclose(sums[row], ref[k], rtol=1e-12)
    np.testing.assert_allclose(sums, c_oracle.metrics(topk, indptr, flat, Ks), rtol=1e-12)


def test_sampler_semantics(dev):
    """rd.sample-unique users, positive from the user's train row with its time slot, negative outside the
    row, popularity gathered at [item, slot of the positive]  (MF/train_new_api.py:366-412)."""
    from pda_amd import ops
    rng = np.random.default_rng(37)
    nU, nI, T, B = 3000, 500, 9, 2048
    rows = [np.sort(rng.permutation(nI)[:rng.integers(0, 60)]).astype(np.int32) for _ in range(nU)]
    rows[5] = np.zeros(0, np.int32)
    indptr = np.zeros(nU + 1, np.int64)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    flat = np.concatenate(rows)
    slots = rng.integers(0, T, len(flat)).astype(np.int32)
    popm = rng.uniform(0, 1, (nI, T)).astype(np.float32)
    ip, ix, sl, pm = to(dev, indptr, flat, slots, popm)
    seen = []
    for step in range(3):
        u, p, n, pp, pn = ops.sample_triplets(ip, ix, B, seed=2020, step=step, n_pool=nU, train_slots=sl,
                                              neg_range=(0, nI), pop_matrix=pm)
        u, p, n, pp, pn = (t.cpu().numpy() for t in (u, p, n, pp, pn))
        assert len(set(u.tolist())) == B and u.min() >= 0 and u.max() < nU
        for r in range(B):
            row = rows[u[r]]
            if len(row) == 0:
                assert p[r] == 0
                continue
            assert p[r] in row and n[r] not in row and 0 <= n[r] < nI
            cand = slots[indptr[u[r]]:indptr[u[r] + 1]][row == p[r]]
            assert any(pp[r] == popm[p[r], s] and pn[r] == popm[n[r], s] for s in cand)
        seen.append(u.copy())
    assert not np.array_equal(seen[0], seen[1])
    u2, p2, n2, _, _ = ops.sample_triplets(ip, ix, B, seed=2020, step=0, n_pool=nU, neg_range=(0, nI))
    np.testing.assert_array_equal(u2.cpu().numpy(), seen[0])     # counter-based: reproducible
    # shard-local negatives (item-parallel training)
    _, _, n3, _, _ = ops.sample_triplets(ip, ix, B, seed=1, step=0, n_pool=nU, neg_range=(100, 200))
    n3 = n3.cpu().numpy()
    assert n3.min() >= 100 and n3.max() < 200


def test_hot_positive_runs_and_batch_sort(dev):
    """Zipf-hot positives: the in-block run combining must give the same update for any batch order, and
    pda_sort_triplets_by_pos must be a pure permutation of the batch (all five arrays together)."""
    from pda_amd import ops
    rng = np.random.default_rng(41)
    nU, nI, d, B, regs, lr = 6000, 300, 64, 2048, 1e-2, 0.05
    U = (rng.standard_normal((nU, d)) * 0.2).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * 0.2).astype(np.float32)
    users = rng.permutation(nU)[:B].astype(np.int32)
    w = 1.0 / np.arange(1, nI + 1)
    pos = rng.choice(nI, size=B, p=w / w.sum()).astype(np.int32)          # the hottest item ~330 times
    neg = rng.integers(0, nI, B).astype(np.int32)
    pp = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
    pn = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
    assert np.bincount(pos).max() > 200
    U1, I1, _, ref_loss = po.train_step(U, I, users, pos, neg, pp, pn, regs, B, lr, optimizer="sgd")
    out = []
    for sort in (False, True):
        Ut, It, ut, pt, nt, ppt, pnt = to(dev, U, I, users, pos, neg, pp, pn)
        if sort:
            ops.sort_triplets_by_pos(ut, pt, nt, ppt, pnt)
            got = np.stack([ut.cpu().numpy(), pt.cpu().numpy(), nt.cpu().numpy()], 1)
            assert np.all(np.diff(got[:, 1]) >= 0)
            ref_rows = {tuple(r) for r in np.stack([users, pos, neg], 1).tolist()}
            assert {tuple(r) for r in got.tolist()} == ref_rows                  # a permutation of whole triplets
            o = np.argsort(got[:, 0]); o2 = np.argsort(users)
            np.testing.assert_array_equal(ppt.cpu().numpy()[o], pp[o2])            # pops moved with their triplets
        loss = torch.zeros(3, device=dev)
        # unsorted batch: PDA_UPD_ANY_ORDER (equal positives combined anywhere in a workgroup); sorted: run combining
        ops.bpr_step(Ut, It, ut, pt, nt, ppt, pnt, regs=regs, reg_div=B, lr=lr, mode=ops.UPD_SGD_FUSED, loss_acc=loss, grouped=sort)
        np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, atol=TOL, rtol=TOL)
        np.testing.assert_allclose(Ut.cpu().numpy(), U1, atol=TOL)
        np.testing.assert_allclose(It.cpu().numpy(), I1, atol=TOL)
        out.append(It.cpu().numpy())
    np.testing.assert_allclose(out[0], out[1], atol=2e-6)


@pytest.mark.parametrize("with_pop", [False, True])
@pytest.mark.parametrize("d", [64, 128])
def test_item_parallel_step_equals_the_fused_step_on_the_concatenated_batch(dev, with_pop, d):
    """SURVEY 8(e) train row: R ranks (emulated one after the other on this GPU), each owning an item slice and a
    sub-batch with positives and negatives inside it; after the exchange every replica of U and the union of the item
    slices must equal one SGD step of the float64 oracle on the concatenated batch, and the loss shares must add up."""
    from pda_amd import dist as pdist
    from pda_amd import ops
    rng = np.random.default_rng(5 + d)
    R, nU, nI, Bl, regs, lr = 4, 6000, 1024, 512, 1e-2, 0.5
    Bg, per = R * Bl, nI // R
    U = (rng.standard_normal((nU, d)) * 0.2).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * 0.2).astype(np.float32)
    users = rng.permutation(nU)[:Bg].astype(np.int32)
    pos = np.concatenate([rng.integers(r * per, r * per + per // 8, Bl) for r in range(R)]).astype(np.int32)   # hot positives
    neg = np.concatenate([rng.integers(r * per, (r + 1) * per, Bl) for r in range(R)]).astype(np.int32)
    pp = (rng.uniform(0, 1, Bg) ** 0.22).astype(np.float32) if with_pop else None
    pn = (rng.uniform(0, 1, Bg) ** 0.22).astype(np.float32) if with_pop else None
    U1, I1, _, ref_loss = po.train_step(U, I, users, pos, neg, pp, pn, regs, Bg, lr, optimizer="sgd")

    Ut = torch.from_numpy(U).to(dev)
    trainers, bufs = [], []
    for r in range(R):
        sl = slice(r * Bl, (r + 1) * Bl)
        shard = torch.from_numpy(I[r * per:(r + 1) * per].copy()).to(dev)
        t = pdist.ItemShardedBPR(Ut, shard, r * per, regs=regs, lr=lr, global_batch=Bg, rank=r, world=R)
        ut, pt, nt, ppt, pnt = to(dev, users[sl], pos[sl], neg[sl], None if pp is None else pp[sl], None if pn is None else pn[sl])
        bufs.append(t.local_step(ut, pt, nt, ppt, pnt))
        trainers.append(t)
    assert torch.equal(Ut.cpu(), torch.from_numpy(U))                  # U untouched before the exchange
    loss = trainers[0].apply(torch.cat(bufs))                          # what every rank does after the all-gather
    np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, atol=TOL, rtol=TOL)
    np.testing.assert_allclose(Ut.cpu().numpy(), U1, atol=TOL)
    got_I = torch.cat([t.I_shard for t in trainers]).cpu().numpy()
    np.testing.assert_allclose(got_I, I1, atol=TOL)
    assert np.abs(got_I - I).max() > 3e-4 and np.abs(Ut.cpu().numpy() - U).max() > 3e-5   # far above TOL: something moved


def test_item_parallel_training_with_shard_samplers_learns(dev):
    """Two emulated ranks, each with its own ShardSampler (positives and negatives inside its slice) and item slice;
    the user replica is shared.  Checks the sampler's slice semantics and that the ranking objective improves."""
    from pda_amd import dist as pdist
    from pda_amd import synthetic
    W = synthetic.make_workload("tiny", dev)
    R, Bl, regs, lr = 2, 256, 1e-3, 10.0      # mean-loss SGD: gradients carry 1/B, hence the large step
    Bg = R * Bl
    U = W.U.clone()
    trainers, samplers = [], []
    for r in range(R):
        lo, hi = pdist.shard_range(W.n_items, r, R)
        trainers.append(pdist.ItemShardedBPR(U, W.I[lo:hi].clone(), lo, regs=regs, lr=lr, global_batch=Bg, rank=r, world=R))
        samplers.append(pdist.ShardSampler(W.hist_indptr, W.hist_indices, lo, hi, Bl, seed=2020, rank=r,
                                           train_slots=W.hist_slots, pop_matrix=W.pop_train))
    ip, ix = W.hist_indptr.cpu().numpy(), W.hist_indices.cpu().numpy()
    losses = []
    for step in range(120):
        bufs = []
        for r in range(R):
            users, pos, neg, pp, pn = samplers[r](step % 4)           # four fixed global batches, revisited
            if step == 0:
                lo, hi = trainers[r].item_offset, trainers[r].item_offset + trainers[r].I_shard.shape[0]
                u, p, n = users.cpu().numpy(), pos.cpu().numpy(), neg.cpu().numpy()
                assert len(set(u.tolist())) == Bl and (p >= lo).all() and (p < hi).all() and (n >= lo).all() and (n < hi).all()
                for a, b, c in zip(u, p, n):
                    row = ix[ip[a]:ip[a + 1]]
                    assert b in row and c not in row
            bufs.append(trainers[r].local_step(users, pos, neg, pp, pn))
        losses.append(float(trainers[0].apply(torch.cat(bufs))[1]))
    assert np.isfinite(losses).all() and np.mean(losses[-4:]) < np.mean(losses[:4]) - 1e-2, (losses[:4], losses[-4:])


@pytest.mark.parametrize("d", [64, 256])
def test_bf16_tables_step(dev, d):
    """pda_bpr_step_bf16: forward on the bf16 rows == the fp32 step on the widened rows (loss, per-occurrence gradients);
    fused SGD moves the fp32 masters by exactly those gradients and pda_refresh_rows_bf16 re-rounds the touched rows."""
    from pda_amd import ops
    rng = np.random.default_rng(70 + d)
    nU, nI, B, regs, lr = 3000, 900, 1024, 1e-2, 0.5
    Um = (rng.standard_normal((nU, d)) * 0.3).astype(np.float32)          # fp32 masters
    Im = (rng.standard_normal((nI, d)) * 0.3).astype(np.float32)
    Ub, Ib = torch.from_numpy(Um).to(dev).bfloat16(), torch.from_numpy(Im).to(dev).bfloat16()
    Uw, Iw = Ub.float().cpu().numpy(), Ib.float().cpu().numpy()           # what the forward pass sees
    users, pos, neg = triplets(rng, nU, nI, B)
    pp = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
    pn = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
    fw = po.bpr_forward(Uw, Iw, users, pos, neg, pp, pn)
    ref_loss = po.bpr_loss(fw, regs, B)
    rdu, rdp, rdn = po.bpr_grads(fw, regs, B, pp, pn)
    ut, pt, nt, ppt, pnt = to(dev, users, pos, neg, pp, pn)
    gu, gp, gn = (torch.empty(B, d, device=dev) for _ in range(3))
    loss = torch.zeros(3, device=dev)
    ops.bpr_step_bf16(Ub, Ib, ut, pt, nt, ppt, pnt, regs=regs, reg_div=B, mode=ops.UPD_NONE, grads_out=(gu, gp, gn), loss_acc=loss)
    np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, atol=TOL, rtol=TOL)
    for g, r in ((gu, rdu), (gp, rdp), (gn, rdn)):
        np.testing.assert_allclose(g.cpu().numpy(), r, atol=TOL)

    Umt, Imt = torch.from_numpy(Um).to(dev), torch.from_numpy(Im).to(dev)
    Ub0, Ib0 = Ub.clone(), Ib.clone()
    loss.zero_()
    ops.bpr_step_bf16(Ub, Ib, ut, pt, nt, ppt, pnt, regs=regs, reg_div=B, lr=lr, mode=ops.UPD_SGD_FUSED, U_master=Umt,
                      I_master=Imt, loss_acc=loss)
    np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, atol=TOL, rtol=TOL)
    U1, I1 = Um.astype(np.float64), Im.astype(np.float64)
    np.subtract.at(U1, users, lr * rdu)
    np.subtract.at(I1, pos, lr * rdp)
    np.subtract.at(I1, neg, lr * rdn)
    np.testing.assert_allclose(Umt.cpu().numpy(), U1, atol=TOL)
    np.testing.assert_allclose(Imt.cpu().numpy(), I1, atol=TOL)
    # touched rows of the bf16 tables = RNE of the masters, bit for bit; untouched rows untouched
    tu, ti = np.unique(users), np.unique(np.concatenate([pos, neg]))
    assert torch.equal(Ub[tu], Umt[tu].bfloat16()) and torch.equal(Ib[ti], Imt[ti].bfloat16())
    mu, mi = np.ones(nU, bool), np.ones(nI, bool)
    mu[tu] = False
    mi[ti] = False
    assert torch.equal(Ub[mu], Ub0[mu]) and torch.equal(Ib[mi], Ib0[mi])
    assert not torch.equal(Ib[ti], Ib0[ti])


def test_device_counter_sampler_draws_the_same_batches(dev):
    """pda_sample_triplets_dev(seed, *step_dev) == pda_sample_triplets(seed, step); pda_counter_add advances the stream."""
    from pda_amd import ops, synthetic
    W = synthetic.make_workload("tiny", dev)
    B = 256
    step_dev = torch.full((1,), 5, dtype=torch.int64, device=dev)
    bufs = (torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev),
            torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev))
    for want_step in (5, 6):
        ops.sample_triplets_into(bufs, W.hist_indptr, W.hist_indices, seed=11, step_dev=step_dev, n_pool=W.n_users,
                                 train_slots=W.hist_slots, neg_range=(0, W.n_items), pop_matrix=W.pop_train)
        ref = ops.sample_triplets(W.hist_indptr, W.hist_indices, B, seed=11, step=want_step, n_pool=W.n_users,
                                  train_slots=W.hist_slots, neg_range=(0, W.n_items), pop_matrix=W.pop_train)
        for a, b in zip(bufs, ref):
            assert torch.equal(a, b)
    assert int(step_dev.item()) == 7
    # two-slot form: the sampler itself stores step + 1 into the other slot (no counter launch)
    two = torch.tensor([20, 0], dtype=torch.int64, device=dev)
    for call, want_step in enumerate((20, 21, 22)):
        ops.sample_triplets_into(bufs, W.hist_indptr, W.hist_indices, seed=11, step_dev=two, n_pool=W.n_users,
                                 train_slots=W.hist_slots, neg_range=(0, W.n_items), pop_matrix=W.pop_train, parity=call & 1)
        ref = ops.sample_triplets(W.hist_indptr, W.hist_indices, B, seed=11, step=want_step, n_pool=W.n_users,
                                  train_slots=W.hist_slots, neg_range=(0, W.n_items), pop_matrix=W.pop_train)
        for a, b in zip(bufs, ref):
            assert torch.equal(a, b)
    assert two.tolist() == [22, 23]


@pytest.mark.parametrize("B", [1, 37, 2048, 4096])
def test_group_triplets_by_pos(dev, B):
    """pda_group_triplets_by_pos: a permutation of whole triplets (all five arrays together) in which every run of equal
    positives is contiguous, deterministic from call to call."""
    from pda_amd import ops
    rng = np.random.default_rng(B)
    nI = 300
    w = 1.0 / np.arange(1, nI + 1)
    users = rng.integers(0, 10 ** 6, B).astype(np.int32)
    pos = rng.choice(nI, size=B, p=w / w.sum()).astype(np.int32) * 7919          # spread ids, hot head
    neg = rng.integers(0, 10 ** 6, B).astype(np.int32)
    pp, pn = rng.uniform(0, 1, B).astype(np.float32), rng.uniform(0, 1, B).astype(np.float32)
    outs = []
    for _ in range(2):
        ut, pt, nt, ppt, pnt = to(dev, users, pos, neg, pp, pn)
        ops.group_triplets_by_pos(ut, pt, nt, ppt, pnt)
        got = [t.cpu().numpy() for t in (ut, pt, nt, ppt, pnt)]
        rows = sorted(zip(*(g.tolist() for g in got)))
        assert rows == sorted(zip(users.tolist(), pos.tolist(), neg.tolist(), pp.tolist(), pn.tolist()))
        p = got[1]
        starts = np.flatnonzero(np.r_[True, p[1:] != p[:-1]])
        assert len(starts) == len(np.unique(p))                                   # one run per distinct positive
        outs.append(got)
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


def test_item_parallel_adam_equals_the_reference_optimiser_on_the_concatenated_batch(dev):
    """Item-parallel training with the REFERENCE's optimiser (TF-1.14 Adam, dense decay): four emulated ranks, each with its
    own replica of U and the Adam state of its item slice; after two global steps every U replica and the union of the
    slices equal two oracle Adam steps on the concatenated batches (dense decay: untouched rows move too)."""
    from pda_amd import dist as pdist
    rng = np.random.default_rng(2024)
    R, nU, nI, d, Bl, regs, lr = 4, 3000, 512, 64, 256, 1e-2, 1e-2
    Bg, per = R * Bl, nI // R
    U = (rng.standard_normal((nU, d)) * 0.2).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * 0.2).astype(np.float32)
    trainers = []
    for r in range(R):
        trainers.append(pdist.ItemShardedBPR(torch.from_numpy(U.copy()).to(dev), torch.from_numpy(I[r * per:(r + 1) * per].copy()).to(dev),
                                             r * per, regs=regs, lr=lr, global_batch=Bg, rank=r, world=R, optimizer="adam"))
    Uref, Iref, state = U, I, None
    for step in range(1, 3):
        users = rng.permutation(nU)[:Bg].astype(np.int32)
        pos = np.concatenate([rng.integers(r * per, r * per + per // 4, Bl) for r in range(R)]).astype(np.int32)
        neg = np.concatenate([rng.integers(r * per, (r + 1) * per, Bl) for r in range(R)]).astype(np.int32)
        pp = (rng.uniform(0, 1, Bg) ** 0.22).astype(np.float32)
        pn = (rng.uniform(0, 1, Bg) ** 0.22).astype(np.float32)
        Uref, Iref, state, ref_loss = po.train_step(Uref, Iref, users, pos, neg, pp, pn, regs, Bg, lr, optimizer="adam", state=state, t=step)
        bufs = []
        for r, t in enumerate(trainers):
            sl = slice(r * Bl, (r + 1) * Bl)
            bufs.append(t.local_step(*to(dev, users[sl], pos[sl], neg[sl], pp[sl], pn[sl])))
        allb = torch.cat(bufs)
        losses = [t.apply(allb) for t in trainers]                       # what every rank does after the all-gather
        np.testing.assert_allclose(losses[0].cpu().numpy(), ref_loss, atol=TOL, rtol=TOL)
        for t in trainers:
            adam_close(t.U.cpu().numpy(), Uref)
        adam_close(torch.cat([t.I_shard for t in trainers]).cpu().numpy(), Iref)
    assert np.abs(Uref - U).max() > 1e-3


@pytest.mark.parametrize("d", [64, 128])
def test_step_with_the_next_batch_sampled_in_the_same_launch(dev, d):
    """pda_bpr_step_sample_f32 == pda_bpr_step_f32 followed by pda_sample_triplets_dev: same tables after every step
    (1e-6: fp32 atomics), bit-identical sampled batches and step counters, also when replayed from a HIP graph."""
    from pda_amd import ops, synthetic
    W = synthetic.make_workload("tiny", dev)
    rng = np.random.default_rng(5 + d)
    B, regs, lr, seed = 512, 1e-2, 0.05, 99
    U0 = torch.from_numpy((rng.standard_normal((W.n_users, d)) * 0.2).astype(np.float32)).to(dev)
    I0 = torch.from_numpy((rng.standard_normal((W.n_items, d)) * 0.2).astype(np.float32)).to(dev)
    kw = dict(n_pool=W.n_users, train_slots=W.hist_slots, neg_range=(0, W.n_items), pop_matrix=W.pop_train)

    def mk():
        return (torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev),
                torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev))

    def run(fused, graph):
        U, I = U0.clone(), I0.clone()
        bufs, loss = [mk(), mk()], torch.zeros(3, device=dev)
        step_dev = torch.tensor([3, 0], dtype=torch.int64, device=dev)
        ops.sample_triplets_into(bufs[0], W.hist_indptr, W.hist_indices, seed=seed, step_dev=step_dev, parity=0, **kw)

        def body(i):
            cur, nxt, par = bufs[i & 1], bufs[(i + 1) & 1], (i + 1) & 1
            if fused:
                ops.bpr_step_and_sample(U, I, *cur, regs=regs, reg_div=B, lr=lr, next_out=nxt, train_indptr=W.hist_indptr,
                                        train_indices=W.hist_indices, seed=seed, step_dev=step_dev, parity=par, loss_acc=loss, **kw)
            else:
                ops.bpr_step(U, I, *cur, regs=regs, reg_div=B, lr=lr, mode=ops.UPD_SGD_FUSED, loss_acc=loss)
                ops.sample_triplets_into(nxt, W.hist_indptr, W.hist_indices, seed=seed, step_dev=step_dev, parity=par, **kw)
        if graph:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                body(0); body(1)                                  # warm-up outside the capture (two calls: parity back to 0)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(4):
                    body(i)
            g.replay()
        else:
            for i in range(6):
                body(i)
        torch.cuda.synchronize()
        return U, I, bufs, step_dev, loss

    ref = run(False, False)
    for fused, graph in ((True, False), (True, True)):
        got = run(fused, graph)
        np.testing.assert_allclose(got[0].cpu().numpy(), ref[0].cpu().numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(got[1].cpu().numpy(), ref[1].cpu().numpy(), rtol=0, atol=2e-6)
        for a, b in zip(got[2][0] + got[2][1], ref[2][0] + ref[2][1]):
            assert torch.equal(a, b)
        assert got[3].tolist() == ref[3].tolist()
        np.testing.assert_allclose(got[4].cpu().numpy(), ref[4].cpu().numpy(), rtol=1e-5)
    # argument checks: the next batch must not land in the buffers this step reads
    U, I = U0.clone(), I0.clone()
    b = mk()
    ops.sample_triplets_into(b, W.hist_indptr, W.hist_indices, seed=seed, step_dev=torch.tensor([0, 0], dtype=torch.int64, device=dev), parity=0, **kw)
    with pytest.raises(ops.PdaHipError if hasattr(ops, "PdaHipError") else Exception):
        ops.bpr_step_and_sample(U, I, *b, regs=regs, reg_div=B, lr=lr, next_out=b, train_indptr=W.hist_indptr, train_indices=W.hist_indices,
                                seed=seed, step_dev=torch.tensor([0, 0], dtype=torch.int64, device=dev), parity=0, **kw)


def test_sampler_many_batches_ahead(dev):
    """pda_sample_batches_dev: 9 batches in one launch == 9 pda_sample_triplets_dev calls (steps c .. c + 8), bit for bit, and
    the step counter moves by 9; with group_by_pos every batch is a permutation of whole triplets of the ungrouped batch with
    the equal positives contiguous."""
    from pda_amd import ops, synthetic
    W = synthetic.make_workload("tiny", dev)
    B, n, seed = 512, 9, 77
    kw = dict(n_pool=W.n_users, train_slots=W.hist_slots, neg_range=(0, W.n_items), pop_matrix=W.pop_train)
    mk = lambda *shape: (torch.empty(shape, dtype=torch.int32, device=dev), torch.empty(shape, dtype=torch.int32, device=dev),
                         torch.empty(shape, dtype=torch.int32, device=dev), torch.empty(shape, device=dev), torch.empty(shape, device=dev))
    ctr = torch.tensor([5], dtype=torch.int64, device=dev)
    singles = []
    for j in range(n):
        b = mk(B)
        ops.sample_triplets_into(b, W.hist_indptr, W.hist_indices, seed=seed, step_dev=ctr, **kw)
        singles.append(b)
    for grouped in (False, True):
        many = mk(n, B)
        sd = torch.tensor([5, 0], dtype=torch.int64, device=dev)
        ops.sample_batches_into(many, W.hist_indptr, W.hist_indices, seed=seed, step_dev=sd, parity=0, group_by_pos=grouped, **kw)
        assert sd.tolist() == [5, 5 + n] and int(ctr) == 5 + n
        for j in range(n):
            got = [t[j] for t in many]
            if not grouped:
                for a, b in zip(got, singles[j]):
                    assert torch.equal(a, b), j
            else:
                key = lambda t5: sorted(zip(*[x.cpu().tolist() for x in t5]))
                assert key(got) == key(singles[j])
                p = got[1].cpu().numpy()
                starts = np.flatnonzero(np.r_[True, p[1:] != p[:-1]])
                assert len(starts) == len(np.unique(p))              # every positive forms ONE run


@pytest.mark.parametrize("d", [64, 128])
def test_training_steps_in_one_launch(dev, d):
    """pda_bpr_train_steps_f32 (a resident grid looping over the steps, the sampler one batch ahead, a grid barrier between
    the steps) == the same number of (pda_bpr_step_f32, pda_sample_triplets_dev) pairs on one stream: tables to 2e-6 (fp32
    atomics) after 1, 2 and 9 steps and after two chained calls, bit-identical batch buffers and step counter, per-step
    losses."""
    from pda_amd import ops, synthetic
    W = synthetic.make_workload("tiny", dev)
    rng = np.random.default_rng(50 + d)
    B, regs, lr, seed = 512, 1e-2, 0.05, 123
    U0 = torch.from_numpy((rng.standard_normal((W.n_users, d)) * 0.2).astype(np.float32)).to(dev)
    I0 = torch.from_numpy((rng.standard_normal((W.n_items, d)) * 0.2).astype(np.float32)).to(dev)
    kw = dict(n_pool=W.n_users, train_slots=W.hist_slots, neg_range=(0, W.n_items), pop_matrix=W.pop_train)

    def mk():
        return (torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev),
                torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev))

    def reference(n):
        U, I, bufs = U0.clone(), I0.clone(), [mk(), mk()]
        ctr = torch.tensor([7], dtype=torch.int64, device=dev)
        ops.sample_triplets_into(bufs[0], W.hist_indptr, W.hist_indices, seed=seed, step_dev=ctr, **kw)      # batch of step 7; ctr -> 8
        losses = torch.zeros((n, 3), device=dev)
        for i in range(n):
            ops.bpr_step(U, I, *bufs[i & 1], regs=regs, reg_div=B, lr=lr, mode=ops.UPD_SGD_FUSED, loss_acc=losses[i])
            ops.sample_triplets_into(bufs[(i + 1) & 1], W.hist_indptr, W.hist_indices, seed=seed, step_dev=ctr, **kw)
        torch.cuda.synchronize()
        return U, I, bufs, ctr, losses

    def looped(chunks):
        U, I, bufs = U0.clone(), I0.clone(), [mk(), mk()]
        ctr = torch.tensor([7], dtype=torch.int64, device=dev)
        ops.sample_triplets_into(bufs[0], W.hist_indptr, W.hist_indices, seed=seed, step_dev=ctr, **kw)
        n = sum(chunks)
        losses = torch.zeros((n, 3), device=dev)
        done = 0
        for c in chunks:
            order = bufs if done % 2 == 0 else bufs[::-1]          # the set that holds the next batch goes first
            _, ws = ops.bpr_train_steps(U, I, order, c, regs=regs, reg_div=B, lr=lr, train_indptr=W.hist_indptr, train_indices=W.hist_indices,
                                        seed=seed, step_ctr=ctr, loss_steps=losses[done:done + c], **kw)
            done += c
            assert int(ws[1]) == 0
        torch.cuda.synchronize()
        return U, I, bufs, ctr, losses

    for chunks in ((1,), (2,), (9,), (3, 4)):
        n = sum(chunks)
        ref, got = reference(n), looped(chunks)
        np.testing.assert_allclose(got[0].cpu().numpy(), ref[0].cpu().numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(got[1].cpu().numpy(), ref[1].cpu().numpy(), rtol=0, atol=2e-6)
        for a, b in zip(got[2][0] + got[2][1], ref[2][0] + ref[2][1]):
            assert torch.equal(a, b), chunks
        assert int(got[3]) == int(ref[3]) == 8 + n
        np.testing.assert_allclose(got[4].cpu().numpy(), ref[4].cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert not torch.equal(ref[0], U0)


def test_training_steps_in_one_launch_large_batch(dev):
    """B = 16 384 at d = 64: 512 step tiles and 256 sampler tiles on a grid of 384 + 64 workgroups -- every workgroup strides
    over several tiles per step.  Per-step losses (a missing tile would lower them by 1/512), the sampled batches and the
    counter equal the launch-per-step sequence; tables to fp32-atomics accuracy."""
    from pda_amd import ops, synthetic
    W = synthetic.make_workload("c2", dev)
    B, regs, lr, seed, n = 16384, 1e-2, 0.05, 321, 3
    kw = dict(n_pool=W.n_users, train_slots=W.hist_slots, neg_range=(0, W.n_items), pop_matrix=W.pop_train)

    def mk():
        return (torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev),
                torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev))
    out = []
    for looped in (False, True):
        U, I, bufs = W.U.clone(), W.I.clone(), [mk(), mk()]
        ctr = torch.tensor([11], dtype=torch.int64, device=dev)
        ops.sample_triplets_into(bufs[0], W.hist_indptr, W.hist_indices, seed=seed, step_dev=ctr, **kw)
        losses = torch.zeros((n, 3), device=dev)
        if looped:
            _, ws = ops.bpr_train_steps(U, I, bufs, n, regs=regs, reg_div=B, lr=lr, train_indptr=W.hist_indptr, train_indices=W.hist_indices,
                                        seed=seed, step_ctr=ctr, loss_steps=losses, **kw)
            assert int(ws[1]) == 0
        else:
            for i in range(n):
                ops.bpr_step(U, I, *bufs[i & 1], regs=regs, reg_div=B, lr=lr, mode=ops.UPD_SGD_FUSED, loss_acc=losses[i])
                ops.sample_triplets_into(bufs[(i + 1) & 1], W.hist_indptr, W.hist_indices, seed=seed, step_dev=ctr, **kw)
        torch.cuda.synchronize()
        out.append((U, I, bufs, int(ctr), losses))
    ref, got = out
    assert got[3] == ref[3] == 12 + n
    for a, b in zip(got[2][0] + got[2][1], ref[2][0] + ref[2][1]):
        assert torch.equal(a, b)
    np.testing.assert_allclose(got[4].cpu().numpy(), ref[4].cpu().numpy(), rtol=1e-4)
    np.testing.assert_allclose(got[0].cpu().numpy(), ref[0].cpu().numpy(), rtol=0, atol=5e-6)
    np.testing.assert_allclose(got[1].cpu().numpy(), ref[1].cpu().numpy(), rtol=0, atol=5e-5)       # hot item rows: hogwild within a step
    assert float((got[1] - W.I).abs().max()) > 1e-8              # (updates are lr x gradient / B: tiny at this batch size)


def test_two_table_adam_sweep_equals_two_sweeps(dev):
    """pda_adam_dense_sweep2_f32 is the arithmetic of two pda_adam_dense_sweep_f32 calls, bit for bit (sizes that do not split
    evenly over the workgroups, sparse gradients, the accumulators zeroed)."""
    from pda_amd import ops
    g = torch.Generator(device=dev); g.manual_seed(3)
    def state(n):
        var = torch.randn(n, generator=g, device=dev)
        m, v = torch.randn(n, generator=g, device=dev) * 0.01, torch.rand(n, generator=g, device=dev) * 0.01
        gr = torch.randn(n, generator=g, device=dev) * (torch.rand(n, generator=g, device=dev) < 0.05)
        return [var, m, v, gr]
    for na, nb in ((50000 * 64, 20000 * 64), (1028, 7 * 4), (3 * 4, 999 * 64)):
        a, b = state(na), state(nb)
        a2, b2 = [t.clone() for t in a], [t.clone() for t in b]
        ops.adam_dense_sweep(*a, 1e-3)
        ops.adam_dense_sweep(*b, 1e-3)
        ops.adam_dense_sweep2(*a2, *b2, 1e-3)
        for x, y in zip(a + b, a2 + b2):
            assert torch.equal(x, y)
        assert float(a2[3].abs().max()) == 0.0 and float(b2[3].abs().max()) == 0.0


@pytest.mark.parametrize("d", [32, 64, 128, 256])
def test_six_stream_adam_sweep_equals_the_seven_stream_sweep(dev, d):
    """pda_adam_mark_rows + pda_adam_dense_sweep3_f32 (the gradient tables read and cleared on the batch's rows only) leave the tables, the moments and the
    gradient accumulators of pda_adam_dense_sweep2_f32, bit for bit, over several steps with repeated rows; the marks are spent after every sweep."""
    from pda_amd import ops
    g = torch.Generator(device=dev); g.manual_seed(11 + d)
    nU, nI, B = 3001, 1777, 512
    def tables():
        return [torch.randn(n, d, generator=g, device=dev) * s for n in (nU, nI) for s in (0.1, 0.01, 0.001)]        # var, m, v(>= 0 below) per table
    a = tables()
    a[2].abs_(); a[5].abs_()
    b = [t.clone() for t in a]
    gUa, gIa = torch.zeros(nU, d, device=dev), torch.zeros(nI, d, device=dev)
    gUb, gIb = gUa.clone(), gIa.clone()
    tu, ti = ops.adam_touched_bitmaps(nU, nI, dev)
    for t in range(1, 5):
        users = torch.randint(0, nU, (B,), generator=g, device=dev, dtype=torch.int32)
        pos = torch.randint(0, 40, (B,), generator=g, device=dev, dtype=torch.int32)                # hot items: many repeats
        neg = torch.randint(0, nI, (B,), generator=g, device=dev, dtype=torch.int32)
        for gU, gI in ((gUa, gIa), (gUb, gIb)):
            gU.index_add_(0, users.long(), torch.ones(B, d, device=dev) * 0.01 * t)
            gI.index_add_(0, pos.long(), torch.ones(B, d, device=dev) * 0.02)
            gI.index_add_(0, neg.long(), torch.ones(B, d, device=dev) * -0.03)
        lr_t = ops.adam_lr_t(1e-3, t)
        ops.adam_dense_sweep2(a[0], a[1], a[2], gUa, a[3], a[4], a[5], gIa, lr_t)
        ops.adam_mark_rows(users, pos, neg, tu, ti)
        assert int(tu.count_nonzero()) > 0
        ops.adam_dense_sweep3(b[0], b[1], b[2], gUb, tu, b[3], b[4], b[5], gIb, ti, lr_t)
        for x, y in zip(a + [gUa, gIa], b + [gUb, gIb]):
            assert torch.equal(x, y)
        assert int(tu.count_nonzero()) == 0 and int(ti.count_nonzero()) == 0 and float(gUb.abs().max()) == 0.0


@pytest.mark.parametrize("d", [32, 64, 128, 256])
@pytest.mark.parametrize("policy", [0, 1, 2])
def test_two_launch_adam_step_equals_the_step_plus_seven_stream_sweep(dev, d, policy):
    """pda_adam_step_f32 (round 6: the step kernel tags the rows it touched, adam_dense_sweep4_kernel reads the gradient tables only there; resident
    or streaming cache policy) leaves the tables, moments, accumulators and loss words of pda_bpr_step_f32(DENSE_GRAD) + pda_adam_dense_sweep2_f32
    bit for bit, over several steps with hot items, with and without popularity, distinct users asserted (plain gU stores) or not."""
    from pda_amd import ops
    g = torch.Generator(device=dev); g.manual_seed(5 + d + policy)
    nU, nI, B, regs, lr = 3001, 1777, 512, 1e-2, 1e-2
    U0, I0 = torch.randn(nU, d, generator=g, device=dev) * 0.1, torch.randn(nI, d, generator=g, device=dev) * 0.1
    z = torch.zeros_like
    Ua, Ia, Ub, Ib = U0.clone(), I0.clone(), U0.clone(), I0.clone()
    sa = [z(Ua), z(Ua), z(Ua), z(Ia), z(Ia), z(Ia)]          # mU vU gU mI vI gI
    sb = [z(Ub), z(Ub), z(Ub), z(Ib), z(Ib), z(Ib)]
    tagU, tagI = ops.adam_row_tags(nU, nI, dev)
    for t in range(1, 7):
        distinct = t % 2 == 0
        users = (torch.randperm(nU, generator=g, device=dev)[:B] if distinct else torch.randint(0, nU, (B,), generator=g, device=dev)).to(torch.int32)
        pos = torch.randint(0, 40 if t % 3 else nI, (B,), generator=g, device=dev, dtype=torch.int32)
        neg = torch.randint(0, nI, (B,), generator=g, device=dev, dtype=torch.int32)
        pp = torch.rand(B, generator=g, device=dev) if t > 2 else None
        pn = torch.rand(B, generator=g, device=dev) if t > 2 else None
        la, lb = torch.zeros(3, device=dev), torch.zeros(3, device=dev)
        lr_t = ops.adam_lr_t(lr, t)
        ops.bpr_step(Ua, Ia, users, pos, neg, pp, pn, regs=regs, reg_div=B, mode=ops.UPD_DENSE_GRAD, gU=sa[2], gI=sa[5], loss_acc=la)
        ops.adam_dense_sweep2(Ua, sa[0], sa[1], sa[2], Ia, sa[3], sa[4], sa[5], lr_t)
        ops.adam_step(Ub, sb[0], sb[1], sb[2], tagU, Ib, sb[3], sb[4], sb[5], tagI, users, pos, neg, pp, pn, regs=regs, reg_div=B, step=t, lr_t=lr_t,
                      users_distinct=distinct, cache_policy=policy, loss_acc=lb)
        # rows referenced once per batch: one gradient term, bit-identical; repeated rows are summed by atomics in either path, in an order that
        # may differ between two launches: equal to rounding of that order
        once_u = torch.bincount(users.long(), minlength=nU) <= 1
        once_i = torch.bincount(torch.cat([pos, neg]).long(), minlength=nI) <= 1
        # (repeated rows: a gradient element that is a cancellation residual of ~1e-10 moves x by lr_t m / (sqrt(v) + eps) ~ 3e-5 |g| / 1e-8 in the first steps --
        # Adam amplifies the order of the atomic sums; TF's own scatter-add has the same freedom.  1e-6 flaked once in eight runs of the GPU suite.)
        for x, y, once in ((Ua, Ub, once_u), (sa[0], sb[0], once_u), (sa[1], sb[1], once_u), (Ia, Ib, once_i), (sa[3], sb[3], once_i), (sa[4], sb[4], once_i)):
            assert torch.equal(x[once], y[once])
            torch.testing.assert_close(x, y, atol=2e-5, rtol=1e-5)
        assert float(sb[2].abs().max()) == 0.0 and float(sb[5].abs().max()) == 0.0
        torch.testing.assert_close(la, lb, atol=1e-6, rtol=1e-5)
        # keep the two paths in step so that rounding differences of the atomic order do not compound
        for x, y in ((Ua, Ub), (sa[0], sb[0]), (sa[1], sb[1]), (Ia, Ib), (sa[3], sb[3]), (sa[4], sb[4])):
            y.copy_(x)
    assert int((tagU == 6).sum()) > 0 and int((tagI == 6).sum()) > 0


def test_tagged_adam_sweep_equals_the_seven_stream_sweep_on_given_gradients(dev):
    """pda_adam_dense_sweep4_f32 alone, gradients and tags from the caller: bit-identical to pda_adam_dense_sweep2_f32; a stale tag on a row whose
    gradient is zero is harmless."""
    from pda_amd import ops
    g = torch.Generator(device=dev); g.manual_seed(77)
    nU, nI, d, B = 2000, 900, 64, 300
    a = [torch.randn(n, d, generator=g, device=dev) * s for n in (nU, nI) for s in (0.1, 0.01, 0.001)]
    a[2].abs_(); a[5].abs_()
    b = [t.clone() for t in a]
    gUa, gIa = torch.zeros(nU, d, device=dev), torch.zeros(nI, d, device=dev)
    gUb, gIb = gUa.clone(), gIa.clone()
    tu, ti = ops.adam_row_tags(nU, nI, dev)
    tu[:50] = 3                                                     # stale tags of a "future" step on rows without gradient
    for t in range(1, 5):
        users = torch.randint(0, nU, (B,), generator=g, device=dev)
        items = torch.randint(0, 60, (B,), generator=g, device=dev)
        for gU, gI in ((gUa, gIa), (gUb, gIb)):
            gU.index_add_(0, users, torch.ones(B, d, device=dev) * 0.01 * t)
            gI.index_add_(0, items, torch.ones(B, d, device=dev) * -0.02)
        tu[users] = t
        ti[items] = t
        lr_t = ops.adam_lr_t(1e-3, t)
        ops.adam_dense_sweep2(a[0], a[1], a[2], gUa, a[3], a[4], a[5], gIa, lr_t)
        ops.adam_dense_sweep4(b[0], b[1], b[2], gUb, tu, b[3], b[4], b[5], gIb, ti, t, lr_t, cache_policy=1 + (t & 1))
        for x, y in zip(a + [gUa, gIa], b + [gUb, gIb]):
            assert torch.equal(x, y)


def test_a_c_caller_runs_the_reference_train_step_with_ctypes_alone(dev):
    """INTEGRATION.md section 2, `adam_step`: pda_adam_step_f32 bound with ctypes only (no pda_amd.ops), argument for argument as the stub there -- the same
    tables, moments, tags and loss words as ops.adam_step on a batch without repeated rows (bit for bit), over two steps."""
    import ctypes as C
    from pda_amd import _lib, ops
    lib = C.CDLL(_lib.LIB_PATH)
    lib.pda_adam_step_f32.restype = C.c_int
    g = torch.Generator(device=dev); g.manual_seed(99)
    nU, nI, d, B, regs, lr = 5000, 3000, 64, 1024, 1e-2, 1e-3
    U0, I0 = torch.randn(nU, d, generator=g, device=dev) * 0.1, torch.randn(nI, d, generator=g, device=dev) * 0.1
    z = torch.zeros_like
    def fresh():
        U, I = U0.clone(), I0.clone()
        return U, I, [z(U), z(U), z(U), z(I), z(I), z(I)], ops.adam_row_tags(nU, nI, dev)
    Ua, Ia, sa, ta = fresh()
    Ub, Ib, sb, tb = fresh()
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for t in (1, 2):
        users = torch.randperm(nU, generator=g, device=dev)[:B].to(torch.int32)
        items = torch.randperm(nI, generator=g, device=dev)[:2 * B].to(torch.int32)
        pos, neg = items[:B].contiguous(), items[B:].contiguous()
        pp, pn = torch.rand(B, generator=g, device=dev), torch.rand(B, generator=g, device=dev)
        la, lb = torch.zeros(3, device=dev), torch.zeros(3, device=dev)
        lr_t = lr * (1 - 0.999 ** t) ** 0.5 / (1 - 0.9 ** t)
        rc = lib.pda_adam_step_f32(p(Ua), p(sa[0]), p(sa[1]), p(sa[2]), p(ta[0]), C.c_size_t(nU), p(Ia), p(sa[3]), p(sa[4]), p(sa[5]), p(ta[1]), C.c_size_t(nI),
                                   p(users), p(pos), p(neg), p(pp), p(pn), B, d, C.c_float(regs), C.c_float(B), t, C.c_float(lr_t), C.c_float(0.9), C.c_float(0.999),
                                   C.c_float(1e-8), 0x100 | 0x200, 0, p(la), stream)
        assert rc == 0
        ops.adam_step(Ub, sb[0], sb[1], sb[2], tb[0], Ib, sb[3], sb[4], sb[5], tb[1], users, pos, neg, pp, pn, regs=regs, reg_div=B, step=t, lr_t=lr_t,
                      users_distinct=True, loss_acc=lb)
        torch.cuda.synchronize()
        for x, y in zip([Ua, Ia] + sa + list(ta), [Ub, Ib] + sb + list(tb)):
            assert torch.equal(x, y)
        torch.testing.assert_close(la, lb, atol=1e-6, rtol=1e-6)
    assert lib.pda_adam_step_f32(None, p(sa[0]), p(sa[1]), p(sa[2]), p(ta[0]), C.c_size_t(nU), p(Ia), p(sa[3]), p(sa[4]), p(sa[5]), p(ta[1]), C.c_size_t(nI),
                                 p(users), p(pos), p(neg), None, None, B, d, C.c_float(regs), C.c_float(B), 3, C.c_float(lr), C.c_float(0.9), C.c_float(0.999),
                                 C.c_float(1e-8), 0, 0, None, stream) == -1          # PDA_ERR_ARG, nothing launched
