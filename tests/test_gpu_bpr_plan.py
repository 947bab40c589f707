"""GPU parity: the plan-driven exact mini-batch SGD step (pda_triplet_plan + pda_bpr_step_plan_f32 / _bf16) vs the float64
oracle (MF/model_api.py:83,102-121: gradients of the whole batch from the tables as they stand, duplicates summed, then applied).
Tolerance 1e-6 on the updated rows of a hot-item batch at lr = 0.05; the step is bit-reproducible."""
import numpy as np
import pytest
import torch

from oracle import pda_oracle as po

pytestmark = pytest.mark.gpu


def to(dev, *xs):
    return [None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in xs]


def hot_batch(rng, nU, nI, B, hot_share=0.3, zipf=True):
    """Distinct users (rd.sample, MF/train_new_api.py:380-381); positives Zipf-distributed with one item carrying `hot_share` of
    the batch; negatives uniform -- some of them equal to positives of other triplets."""
    users = rng.permutation(nU)[:B].astype(np.int32)
    if zipf:
        w = 1.0 / np.arange(1, nI + 1)
        pos = rng.choice(nI, size=B, p=w / w.sum()).astype(np.int32)
    else:
        pos = rng.integers(0, nI, B).astype(np.int32)
    pos[rng.random(B) < hot_share] = 7
    neg = rng.integers(0, nI, B).astype(np.int32)
    neg[:5] = 7                                   # the hot positive is also somebody's negative
    return users, pos, neg


def parse_plan(plan, B):
    raw = plan.cpu().numpy()
    w = raw[: 16 + 4 * (2 * B) + 4 * (2 * B + 2) + 4 * (2 * B)].view(np.int32)
    hdr = w[:4]
    seg_item = w[4:4 + 2 * B]
    seg_start = w[4 + 2 * B:4 + 4 * B + 2]
    entries = w[6 + 4 * B:6 + 6 * B]
    flags = raw[24 + 24 * B:24 + 24 * B + B]
    return hdr, seg_item, seg_start, entries, flags


@pytest.mark.parametrize("B", [1, 37, 2048, 4096, 4097, 20000])
def test_triplet_plan_is_the_sorted_segmentation_of_pos_and_neg(dev, B):
    """(B > 4096: pda_triplet_plan_large, the device-wide sort -- same layout, same checks)"""
    from pda_amd import ops
    rng = np.random.default_rng(B)
    nU, nI = max(9000, B + 1000), 700
    n = 3
    us, ps, ns = [], [], []
    for _ in range(n):
        u, p, q = hot_batch(rng, nU, nI, B, zipf=B > 1)
        us.append(u), ps.append(p), ns.append(q)
    ut, pt, nt = to(dev, np.stack(us), np.stack(ps), np.stack(ns))
    plans = ops.triplet_plan(ut, pt, nt)
    assert plans.shape[0] == n
    for j in range(n):
        hdr, seg_item, seg_start, entries, flags = parse_plan(plans[j], B)
        refs = np.concatenate([ps[j], ns[j]])
        order = np.lexsort((np.arange(2 * B), refs))                    # by (item, index): what the kernel's ranking produces
        items_sorted = refs[order]
        heads = np.r_[True, items_sorted[1:] != items_sorted[:-1]]
        n_seg = int(heads.sum())
        assert tuple(hdr) == (n_seg, 0, 2 * B, B)
        # segments may come in any item order (buckets of a hash), but every segment holds exactly one item's references, ascending
        seen = {}
        for s in range(n_seg):
            a, b = seg_start[s], seg_start[s + 1]
            e = entries[a:b]
            assert b > a and (np.diff(e) > 0).all() and (refs[e] == seg_item[s]).all()
            seen[int(seg_item[s])] = b - a
        assert seg_start[0] == 0 and seg_start[n_seg] == 2 * B
        cnt = np.bincount(refs, minlength=nI)
        assert seen == {int(i): int(c) for i, c in enumerate(cnt) if c}
        np.testing.assert_array_equal(flags, (cnt[ps[j]] == 1).astype(np.uint8) | ((cnt[ns[j]] == 1).astype(np.uint8) << 1))
    # a repeated user is flagged
    us[1][B - 1] = us[1][0]
    if B > 1:
        plans = ops.triplet_plan(*to(dev, np.stack(us), np.stack(ps), np.stack(ns)))
        assert [ops.plan_header(plans[j])[1] for j in range(n)] == [0, 1, 0]


@pytest.mark.parametrize("d", [32, 64, 128, 256])
@pytest.mark.parametrize("with_pop", [False, True])
def test_planned_exact_sgd_equals_the_oracle_on_a_hot_item_batch(dev, d, with_pop):
    from pda_amd import ops
    rng = np.random.default_rng(100 + d + int(with_pop))
    nU, nI, B, regs, lr = 6000, 1500, 2048, 1e-2, 0.05
    U = (rng.standard_normal((nU, d)) * 0.2).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * 0.2).astype(np.float32)
    users, pos, neg = hot_batch(rng, nU, nI, B)
    pp = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32) if with_pop else None
    pn = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32) if with_pop else None
    U1, I1, _, ref_loss = po.train_step(U, I, users, pos, neg, pp, pn, regs, B, lr, optimizer="sgd")
    out = []
    for rep in range(2):
        Ut, It, ut, pt, nt, ppt, pnt = to(dev, U, I, users, pos, neg, pp, pn)
        loss = torch.zeros(3, device=dev)
        plan = ops.triplet_plan(ut, pt, nt)[0]
        ops.bpr_step_plan(Ut, It, ut, pt, nt, ppt, pnt, regs=regs, reg_div=B, lr=lr, plan=plan, loss_acc=loss)
        np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, atol=1e-5, rtol=1e-5)
        np.testing.assert_allclose(Ut.cpu().numpy(), U1, atol=1e-6)
        np.testing.assert_allclose(It.cpu().numpy(), I1, atol=1e-6)
        out.append((Ut.clone(), It.clone()))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])          # no atomics: bit-reproducible
    # rows the batch does not touch are bit-identical to the input
    touched = np.zeros(nI, bool)
    touched[pos] = touched[neg] = True
    assert torch.equal(out[0][1].cpu()[~torch.from_numpy(touched)], torch.from_numpy(I[~touched]))


@pytest.mark.parametrize("B,d", [(8192, 64), (16384, 128), (32768, 32)])
def test_planned_exact_sgd_on_large_batches(dev, B, d):
    """Batches beyond one workgroup's LDS sort (pda_triplet_plan_large): the exact step equals the oracle's mini-batch SGD step on a
    hot-item batch (30 % of the batch on one positive: a segment of thousands of references) and is bit-reproducible."""
    from pda_amd import ops
    rng = np.random.default_rng(7 + B)
    nU, nI, regs, lr = B + 5000, 3000, 1e-2, 0.05
    U = (rng.standard_normal((nU, d)) * 0.2).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * 0.2).astype(np.float32)
    users, pos, neg = hot_batch(rng, nU, nI, B)
    pp = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
    pn = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
    U1, I1, _, ref_loss = po.train_step(U, I, users, pos, neg, pp, pn, regs, B, lr, optimizer="sgd")
    out = []
    for rep in range(2):
        Ut, It, ut, pt, nt, ppt, pnt = to(dev, U, I, users, pos, neg, pp, pn)
        loss = torch.zeros(3, device=dev)
        plan = ops.triplet_plan(ut, pt, nt)[0]
        assert ops.plan_header(plan)[1:] == (0, 2 * B, B)
        ops.bpr_step_plan(Ut, It, ut, pt, nt, ppt, pnt, regs=regs, reg_div=B, lr=lr, plan=plan, loss_acc=loss)
        np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, atol=1e-5, rtol=1e-5)
        np.testing.assert_allclose(Ut.cpu().numpy(), U1, atol=1e-6)
        np.testing.assert_allclose(It.cpu().numpy(), I1, atol=2e-6)       # (the hot row: a sum of ~0.3 B terms)
        out.append((Ut.clone(), It.clone()))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


def test_large_plan_path_gives_the_same_tables_as_the_lds_plan(dev, monkeypatch):
    """The two plan builders order their segments differently (hash buckets vs item id) but sum every segment in the same order:
    the step's tables must agree bit for bit."""
    from pda_amd import ops
    rng = np.random.default_rng(5)
    nU, nI, B, d, regs, lr = 9000, 900, 2048, 64, 1e-2, 0.05
    U = (rng.standard_normal((nU, d)) * 0.2).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * 0.2).astype(np.float32)
    users, pos, neg = hot_batch(rng, nU, nI, B)
    res = []
    for large in (False, True):
        monkeypatch.setattr(ops, "PLAN_LDS_MAX_B", 0 if large else 4096)
        Ut, It, ut, pt, nt = to(dev, U, I, users, pos, neg)
        plan = ops.triplet_plan(ut, pt, nt)[0]
        ops.bpr_step_plan(Ut, It, ut, pt, nt, regs=regs, reg_div=B, lr=lr, plan=plan)
        res.append((Ut.cpu(), It.cpu()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_planned_one_launch_step_plain_stores_on_unshared_rows(dev):
    """exact = 0: user rows and once-referenced item rows are exact (plain stores from the rows the launch gathered); shared item
    rows take atomics like PDA_UPD_SGD_FUSED -- within the hogwild bound of that mode (cross terms, <= lr^2 scale)."""
    from pda_amd import ops
    rng = np.random.default_rng(5)
    nU, nI, d, B, regs, lr = 6000, 1500, 64, 2048, 1e-2, 0.05
    U = (rng.standard_normal((nU, d)) * 0.2).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * 0.2).astype(np.float32)
    users, pos, neg = hot_batch(rng, nU, nI, B)
    pp = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
    pn = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
    U1, I1, _, ref_loss = po.train_step(U, I, users, pos, neg, pp, pn, regs, B, lr, optimizer="sgd")
    Ut, It, ut, pt, nt, ppt, pnt = to(dev, U, I, users, pos, neg, pp, pn)
    loss = torch.zeros(3, device=dev)
    plan = ops.triplet_plan(ut, pt, nt)[0]
    ops.bpr_step_plan(Ut, It, ut, pt, nt, ppt, pnt, regs=regs, reg_div=B, lr=lr, plan=plan, exact=False, loss_acc=loss)
    np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, atol=1e-5, rtol=1e-5)
    cnt = np.bincount(np.concatenate([pos, neg]), minlength=nI)
    once = cnt == 1
    got = It.cpu().numpy()
    np.testing.assert_allclose(got[once], I1[once], atol=1e-6)
    np.testing.assert_allclose(got[~once], I1[~once], atol=1e-4)            # shared rows: hogwild inside the launch
    # user rows: a user's gradient reads its positive / negative rows, which another workgroup's atomics may have moved already
    np.testing.assert_allclose(Ut.cpu().numpy(), U1, atol=1e-4)


@pytest.mark.parametrize("d", [64, 256])
def test_planned_exact_sgd_on_bf16_tables(dev, d):
    """Config 5's table type: forward on the bf16 rows, update of the fp32 masters equal to the oracle step on the widened rows
    (1e-6), touched bf16 rows = RNE of their masters, untouched rows untouched -- in the same two launches."""
    from pda_amd import ops
    rng = np.random.default_rng(d)
    nU, nI, B, regs, lr = 5000, 1200, 2048, 1e-2, 0.05
    Um = (rng.standard_normal((nU, d)) * 0.2).astype(np.float32)
    Im = (rng.standard_normal((nI, d)) * 0.2).astype(np.float32)
    U16 = torch.from_numpy(Um).to(dev).to(torch.bfloat16)
    I16 = torch.from_numpy(Im).to(dev).to(torch.bfloat16)
    Uw, Iw = U16.float().cpu().numpy(), I16.float().cpu().numpy()
    users, pos, neg = hot_batch(rng, nU, nI, B)
    pp = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
    pn = (rng.uniform(0, 1, B) ** 0.22).astype(np.float32)
    # oracle: gradients from the WIDENED rows, applied to the masters
    fw = po.bpr_forward(Uw, Iw, users, pos, neg, pp, pn)
    ref_loss = po.bpr_loss(fw, regs, B)
    du, dp, dn = po.bpr_grads(fw, regs, B, pp, pn)
    U1, I1 = Um.astype(np.float64), Im.astype(np.float64)
    np.subtract.at(U1, users, lr * du)
    np.subtract.at(I1, pos, lr * dp)
    np.subtract.at(I1, neg, lr * dn)
    Ut, It, ut, pt, nt, ppt, pnt = to(dev, Um, Im, users, pos, neg, pp, pn)
    U16b, I16b = U16.clone(), I16.clone()
    loss = torch.zeros(3, device=dev)
    plan = ops.triplet_plan(ut, pt, nt)[0]
    ops.bpr_step_plan(U16, I16, ut, pt, nt, ppt, pnt, regs=regs, reg_div=B, lr=lr, plan=plan, loss_acc=loss, U_master=Ut, I_master=It)
    np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(Ut.cpu().numpy(), U1, atol=1e-6)
    np.testing.assert_allclose(It.cpu().numpy(), I1, atol=1e-6)
    tu = torch.zeros(nU, dtype=torch.bool)
    tu[torch.from_numpy(users).long()] = True
    ti = torch.zeros(nI, dtype=torch.bool)
    ti[torch.from_numpy(np.concatenate([pos, neg])).long()] = True
    assert torch.equal(U16.cpu()[tu], Ut.cpu()[tu].to(torch.bfloat16)) and torch.equal(I16.cpu()[ti], It.cpu()[ti].to(torch.bfloat16))
    assert torch.equal(U16.cpu()[~tu], U16b.cpu()[~tu]) and torch.equal(I16.cpu()[~ti], I16b.cpu()[~ti])


def test_planned_step_rejects_a_batch_with_a_repeated_user(dev):
    from pda_amd import ops
    rng = np.random.default_rng(9)
    nU, nI, d, B = 3000, 500, 64, 512
    U = (rng.standard_normal((nU, d)) * 0.2).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * 0.2).astype(np.float32)
    users, pos, neg = hot_batch(rng, nU, nI, B)
    users[100] = users[3]
    Ut, It, ut, pt, nt = to(dev, U, I, users, pos, neg)
    loss = torch.zeros(3, device=dev)
    plan = ops.triplet_plan(ut, pt, nt)[0]
    assert ops.plan_header(plan)[1] == 1
    ops.bpr_step_plan(Ut, It, ut, pt, nt, regs=1e-2, reg_div=B, lr=0.05, plan=plan, loss_acc=loss)
    assert torch.isnan(loss).all()
    assert torch.equal(Ut.cpu(), torch.from_numpy(U)) and torch.equal(It.cpu(), torch.from_numpy(I))


def test_planned_steps_train_like_the_exact_two_launch_path(dev):
    """Twenty steps on device-sampled batches (plans of 8 batches per launch, as the trainer's sampler does): the tables follow
    the existing exact path (pda_bpr_step_f32(PDA_UPD_NONE) + pda_sgd_apply_f32) to 2e-6."""
    from pda_amd import ops, synthetic
    W = synthetic.make_workload("tiny", dev)
    B, regs, lr = 1024, 1e-2, 0.05
    Ua, Ia = W.U.clone(), W.I.clone()
    Ub, Ib = W.U.clone(), W.I.clone()
    la, lb = torch.zeros(3, device=dev), torch.zeros(3, device=dev)
    sc = scp = None
    for s0 in range(0, 24, 8):
        bs = [ops.sample_triplets(W.hist_indptr, W.hist_indices, B, seed=3, step=s0 + j, n_pool=W.n_users, train_slots=W.hist_slots,
                                  neg_range=(0, W.n_items), pop_matrix=W.pop_train) for j in range(8)]
        stack = [torch.stack([b[k] for b in bs]) for k in range(5)]
        plans = ops.triplet_plan(stack[0], stack[1], stack[2])
        for j in range(8):
            b = tuple(x[j] for x in stack)
            scp = ops.bpr_step_plan(Ua, Ia, *b, regs=regs, reg_div=B, lr=lr, plan=plans[j], scratch=scp, loss_acc=la)
            sc = ops.sgd_step_exact(Ub, Ib, *b, regs=regs, reg_div=B, lr=lr, loss_acc=lb, scratch=sc)
    torch.cuda.synchronize()
    np.testing.assert_allclose(la.cpu().numpy(), lb.cpu().numpy(), rtol=1e-5)
    np.testing.assert_allclose(Ua.cpu().numpy(), Ub.cpu().numpy(), atol=2e-6)
    np.testing.assert_allclose(Ia.cpu().numpy(), Ib.cpu().numpy(), atol=2e-6)
