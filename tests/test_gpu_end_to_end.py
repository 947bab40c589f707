"""GPU: the drop-in trainer end to end on a dataset written in the reference's on-disk formats -- CLI, loaders,
device sampler / host generator, fused step + Adam, evaluation driver + device metrics, checkpoints, log lines."""
import os

import numpy as np
import pytest
import torch

from oracle import pda_oracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def toy(tmp_path_factory):
    from pda_amd import synthetic
    root = tmp_path_factory.mktemp("data")
    synthetic.write_dataset(str(root / "toy"), n_users=400, n_items=300, mean_hist=20)
    return str(root) + "/"


def _argv(toy, save, train, extra=()):
    return ["--data_path", toy, "--dataset", "toy", "--train", train, "--test", train, "--epoch", "3", "--log_interval", "2",
            "--batch_size", "256", "--lr", "1e-2", "--regs", "1e-2", "--valid_set", "valid", "--pop_exp", "0.22",
            "--save_dir", save, "--Ks", "[20,50]", "--save_flag", "0", "--saveID", "t", "--cuda", "0", "--eval_block", "128"] + list(extra)


@pytest.mark.parametrize("train,extra", [("s_condition", ()), ("normal", ("--sampler", "host")),
                                         ("s_condition", ("--optimizer", "sgd", "--lr", "0.05")),
                                         ("s_condition", ("--optimizer", "lazy_adam")),
                                         ("s_condition", ("--adam_sweep", "replay")),                    # exact dense decay without the sweep
                                         ("s_condition", ("--adam_sweep", "replay", "--table_dtype", "bf16")),
                                         ("s_condition", ("--table_dtype", "bf16")),                     # bf16 tables, faithful Adam on the masters
                                         ("normal", ("--table_dtype", "bf16", "--optimizer", "sgd", "--lr", "0.05"))])
def test_main_runs_like_the_reference_script(dev, toy, tmp_path, capsys, train, extra):
    from pda_amd import train_new_api as t
    cfg, cfg_main = t.main(_argv(toy, str(tmp_path) + "/", train, extra))
    out = capsys.readouterr().out
    assert "Epoch 0 [" in out and "train==[" in out and "recall=[" in out and "training and testing end!!!!" in out
    assert ("injecting last stage popularity" in out) == (train == "s_condition")
    assert ("best expo" in out) == (train == "normal")
    ck = [f for _, _, fs in os.walk(tmp_path) for f in fs]
    assert "best_ckpt.ckpt" in ck and "best_main_ckpt.ckpt" in ck
    assert 0.0 <= cfg["best_recall"] <= 1.0 and cfg_main["best_recall"] >= 0.0


def test_training_reduces_the_loss_and_beats_random_ranking(dev, toy, tmp_path):
    from pda_amd import train_new_api as t
    from pda_amd.sampler import DeviceSampler
    t.configure(_argv(toy, str(tmp_path) + "/", "s_condition"))
    a, d = t.args, t.data
    pop_all = t.load_popularity(a)
    d.add_expo_popularity(np.power(t.get_popularity_from_load(pop_all), a.pop_exp))
    model = t.DatasetApi_Model(a, {"n_users": d.n_users, "n_items": d.n_items}, 256, DeviceSampler(d, dev, True), dev)
    sess = t.Session(model)
    rec = model.Recommender
    fetches = [rec.opt_pop_global, rec.loss_pop_global, rec.mf_loss_pop_global, rec.reg_loss_pop_global]
    ev = t.evaluation(d, [20, 50], dev, block=128)
    ev.set_evaluate_obj_pre("valid")
    ev.set_testing_popularity(None)
    before = ev.eval(model, sess, "main_branch")
    losses = []
    for epoch in range(30):
        model.switch_to_training_or_reinitsampler(sess)
        tot, n = 0.0, 0
        try:
            while True:
                _, loss, mf, reg = sess.run(fetches)          # the reference's per-step fetch (one sync per step)
                assert abs(loss - (mf + reg)) < 1e-5
                tot, n = tot + loss, n + 1
        except t.OutOfRangeError:
            pass
        assert n == d.n_train // a.batch_size + 1
        losses.append(tot / n)
    assert losses[-1] < losses[0] - 0.02
    after = ev.eval(model, sess, "main_branch")
    assert after["recall"][1] > before["recall"][1] + 0.02      # popularity-skewed data: BPR learns it quickly

    # evaluation driver == oracle on the trained weights (device metrics + fused top-K vs numpy float64)
    U = rec.weights["user_embedding"].cpu().numpy()
    I = rec.weights["item_embedding"].cpu().numpy()
    users = list(d.valid_user_list.keys())
    hist = [sorted(d.train_user_list[u]) for u in users]
    ip = np.zeros(len(users) + 1, np.int64)
    ip[1:] = np.cumsum([len(h) for h in hist])
    pop_last = np.power(pop_all[:, -2], a.pop_exp)
    for rec_type, pop in (("main_branch", None), ("condition", pop_last)):
        ev.set_testing_popularity(pop)
        got = ev.eval(model, sess, rec_type)
        ridx, _ = po.recommend_topk(U, I, np.asarray(users), ip, np.concatenate(hist), 50, rec_type, pop)
        ref = po.evaluate_topk(ridx, users, d.valid_user_list, [20, 50])
        for k in ref:
            np.testing.assert_allclose(got[k], ref[k], atol=2e-3, err_msg=rec_type + " " + k)   # fp32 near-tie swaps only

    # reference-style call: python lists + the (index, [-inf]*nnz, shape) mask triple
    blocks = po.build_eval_blocks(d.valid_user_list, d.train_user_list, block=2048)
    bu, index, rows, nnz = blocks[0]
    mask = (index, np.array([-np.inf] * nnz, dtype=np.float32), np.array([rows, d.n_items], dtype=np.int64))
    topk = model.do_recommendation(sess, bu, list(range(d.n_items)), "condition", pos_pop=pop_last, sparse_cliked_matrix=mask)
    assert topk.shape == (len(bu), 50) and topk.dtype == np.int32
    agree = (topk == ridx[:len(bu)]).mean()
    assert agree > 0.99
    # the next evaluation epoch hands over the SAME mask array: its CSR is taken from the device cache, the result is the same; a caller
    # that refills the array in place (here: the first user's first train item swapped for another one) gets a fresh conversion
    cached = model._mask_cache[id(index)][2]
    again = model.do_recommendation(sess, bu, list(range(d.n_items)), "condition", pos_pop=pop_last, sparse_cliked_matrix=mask)
    assert np.array_equal(again, topk) and model._mask_cache[id(index)][2] is cached
    row0 = np.nonzero(index[:, 0] == index[0, 0])[0]
    freed = int(index[0, 1])
    index[0, 1] = next(i for i in range(d.n_items) if i not in set(index[row0, 1].tolist()) and i not in set(topk[index[0, 0]].tolist()))
    changed = model.do_recommendation(sess, bu, list(range(d.n_items)), "condition", pos_pop=pop_last, sparse_cliked_matrix=mask)
    assert model._mask_cache[id(index)][2] is not cached
    assert np.array_equal(changed[1:], topk[1:]) or index[0, 0] != 0          # only the first user's mask changed
    index[0, 1] = freed
    with pytest.raises(NotImplementedError):
        model.do_recommendation(sess, bu, None, "bogus")
    # dense compatibility surface
    sc = model.testing(sess, bu[:4], list(range(d.n_items)), "condition", pos_pop=pop_last)
    ref_sc = po.score_matrix(U, I, np.asarray(bu[:4]), "condition", pop_last)
    np.testing.assert_allclose(sc, ref_sc, rtol=1e-4, atol=1e-5)

    # checkpoint round trip
    sd = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in rec.state_dict().items()}
    rec.weights["user_embedding"].zero_()
    rec.load_state_dict(sd)
    assert torch.equal(rec.weights["user_embedding"], sd["user_embedding"])


@pytest.mark.parametrize("with_pop", [True, False])
def test_device_sampler_queue_is_the_per_step_sampler(dev, toy, tmp_path, with_pop):
    """DeviceSampler(ahead=32) -- one sampler launch per 32 batches -- hands out, batch for batch, what ahead=1 (one launch per
    step) draws, across refills and across an epoch boundary; a batch stays valid while the next 32 are taken."""
    from pda_amd import train_new_api as t
    from pda_amd.sampler import DeviceSampler
    t.configure(_argv(toy, str(tmp_path) + "/", "s_condition" if with_pop else "normal"))
    a, d = t.args, t.data
    if with_pop:
        d.add_expo_popularity(np.power(t.get_popularity_from_load(t.load_popularity(a)), a.pop_exp))
    one, many = DeviceSampler(d, dev, with_pop, ahead=1), DeviceSampler(d, dev, with_pop, ahead=32)
    kept = []
    for i in range(75):
        x, y = one.batch(), many.batch()
        assert len(x) == len(y) == (5 if with_pop else 3)
        for p, q in zip(x, y):
            assert torch.equal(p, q), i
        kept.append(([p.clone() for p in y], y))
    for clone, view in kept[-32:]:
        for p, q in zip(clone, view):
            assert torch.equal(p, q)
    assert sum(1 for _ in many()) == d.n_train // d.batch_size + 1


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_adam_without_the_sweep_trains_the_same_model(dev, toy, tmp_path, dtype):
    """--adam_sweep replay vs sweep through the drop-in trainer: same seed, same device-sampled batches, two epochs; the tables
    the evaluation reads (after the implicit sync) and the moments in the checkpoint agree to rounding (item rows repeat
    inside a batch: their atomic gradient sums differ in order between two runs), the recommendations agree."""
    from pda_amd import train_new_api as t
    from pda_amd.sampler import DeviceSampler
    res = []
    for mode in ("sweep", "replay"):
        t.configure(_argv(toy, str(tmp_path) + "/", "s_condition", ("--adam_sweep", mode, "--table_dtype", dtype)))
        a, d = t.args, t.data
        pop_all = t.load_popularity(a)
        d.add_expo_popularity(np.power(t.get_popularity_from_load(pop_all), a.pop_exp))
        model = t.DatasetApi_Model(a, {"n_users": d.n_users, "n_items": d.n_items}, 256, DeviceSampler(d, dev, True), dev)
        sess = t.Session(model)
        rec = model.Recommender
        assert rec.adam_exact_lazy == (mode == "replay")
        for epoch in range(2):
            model.switch_to_training_or_reinitsampler(sess)
            try:
                while True:
                    sess.run([rec.opt_pop_global, rec.loss_pop_global])
            except t.OutOfRangeError:
                pass
        if mode == "replay":
            assert int((rec._lazy.lastU < rec._t).sum()) > 0          # rows are behind until something reads the tables
        U, I = (x.float().clone() for x in rec.score_tables())
        sd = rec.state_dict()
        res.append((U, I, sd["mU"].clone(), sd["vI"].clone(), sd["adam_t"]))
        if mode == "replay":
            assert int(rec._lazy.lastU.min()) == rec._t and int(rec._lazy.lastI.min()) == rec._t
    (U0, I0, m0, v0, t0), (U1, I1, m1, v1, t1) = res
    assert t0 == t1 and t0 > 10
    # (two runs of atomics in different orders: elements whose summed gradient is of the size of Adam's epsilon may differ by up
    # to 1e-4 -- a handful; everything else to rounding)
    for x, y in ((U0, U1), (I0, I1), (m0, m1), (v0, v1)):
        err = (x.float() - y.float()).abs()
        if dtype == "bf16":
            assert float(err.max()) <= 2e-2
        else:
            bad = err > (1e-5 + 1e-4 * y.float().abs())
            assert float(bad.float().mean()) < 1e-3 and float(err.max()) < 5e-4, (float(bad.float().mean()), float(err.max()))


def test_eval_user_without_train_rows_raises_keyerror_under_data2(dev, toy, tmp_path):
    from pda_amd import train_new_api as t
    t.configure(_argv(toy, str(tmp_path) + "/", "s_condition"))
    t.data.valid_user_list[10 ** 6] = [1]
    ev = t.evaluation(t.data, [20], dev)
    with pytest.raises(KeyError):
        ev.set_evaluate_obj_pre("valid")


def test_exact_sgd_through_the_trainer_takes_the_planned_step(dev, toy, tmp_path):
    """--optimizer sgd with the device sampler: every batch arrives with its pda_triplet_plan and the step runs without atomics
    (ops.bpr_step_plan); the tables after 20 steps equal those of the plan-less exact path on the same batches (2e-6)."""
    from pda_amd import train_new_api as t, ops
    from pda_amd.sampler import DeviceSampler
    res = []
    for planned in (True, False):
        t.configure(_argv(toy, str(tmp_path) + "/", "s_condition", ("--optimizer", "sgd", "--lr", "0.05")))
        a, d = t.args, t.data
        pop_all = t.load_popularity(a)
        d.add_expo_popularity(np.power(t.get_popularity_from_load(pop_all), a.pop_exp))
        smp = DeviceSampler(d, dev, True)
        model = t.DatasetApi_Model(a, {"n_users": d.n_users, "n_items": d.n_items}, 256, smp, dev)
        assert smp.with_plan is True
        if not planned:
            smp.with_plan = False
        sess = t.Session(model)
        rec = model.Recommender
        fetches = [rec.opt_pop_global, rec.loss_pop_global, rec.mf_loss_pop_global, rec.reg_loss_pop_global]
        model.switch_to_training_or_reinitsampler(sess)
        for _ in range(20):
            sess.run_async(fetches)
            assert (model.batch_plan is not None) == planned
        torch.cuda.synchronize()
        res.append((rec.weights["user_embedding"].clone(), rec.weights["item_embedding"].clone()))
    np.testing.assert_allclose(res[0][0].cpu().numpy(), res[1][0].cpu().numpy(), atol=2e-6)
    np.testing.assert_allclose(res[0][1].cpu().numpy(), res[1][1].cpu().numpy(), atol=2e-6)


@pytest.mark.parametrize("train", ["s_condition", "normal"])
def test_testing_and_predict_return_the_scores_the_recommender_ranks_by(dev, toy, tmp_path, train):
    """DatasetApi_Model.testing / predict (MF/train_new_api.py:642-696: batch_ratings / condition_ratings as a matrix, the NeuRec evaluators'
    protocol) through pda_score_dense_f32: float32 [B, len(items)], equal to the oracle's scores to 1e-5, and -- the same fmaf chain --
    BIT FOR BIT the values do_recommendation's top-K lists carry for the same pairs; a subset / permutation of the items gathers columns."""
    from pda_amd import ops, train_new_api as t
    t.configure(_argv(toy, str(tmp_path) + "/", train))
    args, data = t.args, t.data
    model = t.DatasetApi_Model(args, {"n_users": data.n_users, "n_items": data.n_items}, 256, None, dev)
    sess = t.Session(model)
    model.set_sess(sess)
    rng = np.random.default_rng(5)
    users = rng.permutation(data.n_users)[:37].tolist()
    items = list(range(data.n_items))
    pop = (rng.uniform(0, 1, data.n_items) ** 0.22).astype(np.float32)
    U = model.Recommender.weights["user_embedding"].cpu().numpy().astype(np.float64)
    I = model.Recommender.weights["item_embedding"].cpu().numpy().astype(np.float64)
    R = U[users] @ I.T
    raw = model.testing(sess, users, items, "main_branch")
    assert raw.dtype == np.float32 and raw.shape == (37, data.n_items)
    np.testing.assert_allclose(raw, R, atol=1e-5, rtol=1e-5)
    cond = model.testing(sess, users, items, "condition", pos_pop=pop)
    np.testing.assert_allclose(cond, np.where(R > 0, R + 1.0, np.exp(R)) * pop[None, :], atol=1e-5, rtol=1e-5)
    sub = rng.permutation(data.n_items)[:50].tolist()
    np.testing.assert_array_equal(model.testing(sess, users, sub, "main_branch"), raw[:, sub])
    np.testing.assert_array_equal(model.testing(sess, users, sub, "condition", pos_pop=pop[sub]), cond[:, sub])
    # the lists of the recommender: same pairs, same bits
    ut = torch.as_tensor(np.asarray(users, dtype=np.int32), device=dev)
    Ut, It = model.Recommender.score_tables()
    for head, mat, p in ((ops.HEAD_RAW, raw, None), (ops.HEAD_POP, cond, torch.as_tensor(pop, device=dev))):
        idx, val = ops.recommend_topk(Ut, It, ut, 50, head, p, None)
        idx, val = idx.cpu().numpy(), val.cpu().numpy()
        np.testing.assert_array_equal(val, np.take_along_axis(mat, idx.astype(np.int64), axis=1))
    # predict(): the NeuRec protocol on top of it (:683-696)
    model.set_testing_way("o", None)
    np.testing.assert_array_equal(model.predict(users, None), raw)
    model.set_testing_way("condition", pop)
    np.testing.assert_array_equal(model.predict(users, sub), cond[:, sub])
    with pytest.raises(NotImplementedError):
        model.testing(sess, users, items, "nonsense")
