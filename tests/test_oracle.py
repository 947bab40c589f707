"""CPU suite, part 1: pin the oracle (and the host-side glue) against golden vectors produced by RUNNING the
reference's runnable pieces (tests/golden/make_golden.py), and against an independent second derivation
(torch.autograd, a scatter-style Adam) for the TF-graph arithmetic the reference cannot execute here."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import pda_oracle as po

G = os.path.join(os.path.dirname(__file__), "golden")


def jload(name):
    return json.load(open(os.path.join(G, name)))


# ------------------------------------------------------------------------------------------------ metrics (A8)
@pytest.mark.parametrize("impl", ["oracle", "product_host"])
def test_get_performance_matches_reference_golden(impl):
    if impl == "oracle":
        fn = po.get_performance
    else:
        from pda_amd.used_metric import get_performance as fn
    for c in jload("metrics.json"):
        out = fn(c["target"], c["r"], c["Ks"])
        for k in ("recall", "precision", "ndcg", "hit_ratio"):
            np.testing.assert_allclose(out[k], c["out"][k], rtol=1e-12, atol=0, err_msg=k)


def test_c_metrics_match_reference_golden():
    cases = [c for c in jload("metrics.json") if len(c["r"]) == 50]
    for Ks in ([20, 50], [1, 5, 10, 50]):
        sub = [c for c in cases if c["Ks"] == Ks]
        topk = np.array([c["r"] for c in sub], dtype=np.int32)
        indptr = np.zeros(len(sub) + 1, np.int64)
        indptr[1:] = np.cumsum([len(c["target"]) for c in sub])
        flat = np.concatenate([np.asarray(c["target"], np.int32) for c in sub])
        sums = c_oracle.metrics(topk, indptr, flat, Ks)
        for row, k in enumerate(("precision", "recall", "ndcg", "hit_ratio")):
            ref = np.sum([c["out"][k] for c in sub], axis=0)
            np.testing.assert_allclose(sums[row], ref, rtol=1e-12)


# ------------------------------------------------------------------------------------------------ popularity (N2)
def _stage_counts(files, T):
    stages = []
    for t in range(T):
        rows = [(int(l.split()[0]), len(l.split()) - 1) for l in files["t_%d.txt" % t].splitlines() if l.strip()]
        stages.append(rows)
    return stages


def test_pop_pre_matches_reference_output_file(tmp_path):
    g = jload("pop_pre.json")
    ref = np.array([[float(x) for x in l.split()[1:]] for l in g["item_pop_seq_ori2"].splitlines()])
    np.testing.assert_allclose(po.pop_pre(_stage_counts(g["files"], g["slot_count"])).T, ref, rtol=1e-15)
    # product: same CLI, byte-identical output file
    from pda_amd import pop_pre
    for k, v in g["files"].items():
        (tmp_path / k).write_text(v)
    pop_pre.main(["--path", str(tmp_path) + "/", "--slot_count", str(g["slot_count"])])
    assert (tmp_path / "item_pop_seq_ori2.txt").read_text() == g["item_pop_seq_ori2"]
    assert ref.min(axis=0).max() == 0.0 and ref.max(axis=0).min() == 1.0      # per-slot min exactly 0, max exactly 1


def test_popularity_heads_match_reference_expressions():
    g = np.load(os.path.join(G, "popularity_heads.npz"))
    last, lin, train = po.popularity_heads(g["pop_item_all"].copy(), float(g["gamma"]))
    np.testing.assert_array_equal(last, g["last"])
    np.testing.assert_array_equal(lin, g["linear"])
    np.testing.assert_array_equal(train, g["train"])


# ------------------------------------------------------------------------------------------------ loaders (N2)
def test_data_loader_matches_reference(tmp_path):
    from pda_amd import load_data, parse
    g = jload("loader.json")
    d = tmp_path / "toy"
    d.mkdir()
    for k, v in g["files"].items():
        (d / k).write_text(v)
    a = parse.parse_args(["--data_path", str(tmp_path) + "/", "--dataset", "toy", "--batch_size", "8"])
    D = load_data.Data(a)
    e = g["expected"]
    assert (D.n_users, D.n_items, D.n_train, D.n_valid, D.n_test) == (e["n_users"], e["n_items"], e["n_train"], e["n_valid"], e["n_test"])
    assert {str(k): v for k, v in D.train_user_list.items()} == e["train"]
    assert {str(k): v for k, v in D.valid_user_list.items()} == e["valid"]
    assert {str(k): v for k, v in D.test_user_list.items()} == e["test"]
    assert list(D.train_user_list.keys()) == [int(k) for k in e["train"].keys()]       # file order kept
    assert len(D.items) == e["n_items_list"] and D.batch_size == e["batch_size"]


def test_flags_match_reference_parse():
    from pda_amd import parse
    ref = jload("flags.json")
    mine = vars(parse.parse_args([]))
    for k, v in ref.items():
        assert k in mine, "missing reference flag --" + k
        assert mine[k] == v, (k, mine[k], v)
    # the README commands parse, including the dead --start/--end/--step and the '--test s_condtion' typo
    a = parse.parse_args("--dataset douban --epoch 2000 --save_flag 0 --log_interval 5 --start 0 --end 10 --step 1 --batch_size 2048 "
                         "--lr 1e-2 --train s_condition --test s_condtion --saveID xxx --cuda 0 --regs 1e-2 --valid_set valid "
                         "--pop_exp 0.22 --save_dir /tmp/x/ --Ks [20,50]".split())
    assert a.batch_size == 2048 and eval(a.Ks) == [20, 50] and a.pop_exp == 0.22


# ------------------------------------------------------------------------------------------------ top-K (A6)
def test_topk_matches_reference_native_arg_topk():
    g = np.load(os.path.join(G, "topk_ref.npz"))
    sc = g["scores"]
    idx = po.topk_desc_lower_index_first(sc.astype(np.float64), 50)
    np.testing.assert_array_equal(idx, g["arg_topk"])                    # tie-free rows: lists are identical
    # and, when oracle/_ref is present (authoring container / snapshot), against a live call
    live = c_oracle.ref_arg_topk(sc, 50)
    if live is not None:
        np.testing.assert_array_equal(live, g["arg_topk"])
    # reference cpp_evaluate_matrix (cumulative-per-rank precision, recall, ndcg) at K = 20 and 50
    for u in range(sc.shape[0]):
        tgt = g["tgt_indices"][g["tgt_indptr"][u]:g["tgt_indptr"][u + 1]].tolist()
        out = po.get_performance(tgt, idx[u], [20, 50])
        for q, K in enumerate((20, 50)):
            np.testing.assert_allclose(out["precision"][q], g["eval_matrix"][u, 0, K - 1], rtol=1e-6)
            np.testing.assert_allclose(out["recall"][q], g["eval_matrix"][u, 1, K - 1], rtol=1e-6)
            np.testing.assert_allclose(out["ndcg"][q], g["eval_matrix"][u, 2, K - 1], rtol=2e-6)


def _case(rng, nU=50, nI=300, d=64):
    U = (rng.standard_normal((nU, d)) * 0.1).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * 0.1).astype(np.float32)
    pop = (rng.uniform(0, 1, nI) ** 0.22).astype(np.float32)
    rows = [np.sort(rng.integers(0, nI, rng.integers(0, 30))).astype(np.int32) for _ in range(nU)]
    indptr = np.zeros(nU + 1, np.int64)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    return U, I, pop, indptr, np.concatenate(rows).astype(np.int32)


@pytest.mark.parametrize("head", [0, 1])
def test_c_oracle_equals_numpy_oracle(head):
    rng = np.random.default_rng(3 + head)
    U, I, pop, indptr, indices = _case(rng)
    users = rng.permutation(50).astype(np.int32)
    ip = np.zeros(51, np.int64)
    rows = [indices[indptr[u]:indptr[u + 1]] for u in users]
    ip[1:] = np.cumsum([len(r) for r in rows])
    ix = np.concatenate(rows).astype(np.int32)
    rt = ("main_branch", "condition")[head]
    nidx, nval = po.recommend_topk(U, I, users, ip, ix, 50, rt, pop)                      # float64
    for order in (0, 1):
        cidx, cval = c_oracle.score_topk(U, I, users, 50, head, pop, ip, ix, order=order)
        np.testing.assert_allclose(cval, nval, rtol=2e-6, atol=2e-7)
        assert (cidx == nidx).mean() > 0.995                                              # only fp32 near-ties may swap
    # item shards + merge == whole catalogue
    parts_v, parts_i = [], []
    for lo, hi in ((0, 96), (96, 224), (224, 300)):
        ci, cv = c_oracle.score_topk(U, I, users, 50, head, pop, ip, ix, item_offset=lo, n_items_local=hi - lo, order=1)
        parts_i.append(ci)
        parts_v.append(cv)
    mi, mv = po.merge_partial_topk(np.stack(parts_v), np.stack(parts_i), 50)
    ci, cv = c_oracle.score_topk(U, I, users, 50, head, pop, ip, ix, order=1)
    np.testing.assert_array_equal(mi, ci)
    np.testing.assert_array_equal(mv, cv)


def test_tie_break_is_lower_index_first():
    R = np.array([[1.0, 3.0, 3.0, -np.inf, 3.0, 0.0, -np.inf]])
    np.testing.assert_array_equal(po.topk_desc_lower_index_first(R, 6), [[1, 2, 4, 0, 5, 3]])
    with pytest.raises(ValueError):
        po.topk_desc_lower_index_first(R, 8)


# ------------------------------------------------------------------------------------------------ train step (A1-A5)
@pytest.mark.parametrize("with_pop", [False, True])
def test_closed_form_gradients_equal_autograd(with_pop):
    rng = np.random.default_rng(5)
    nU, nI, d, B, regs = 40, 30, 16, 64, 1e-2
    U = rng.standard_normal((nU, d)) * 0.4
    I = rng.standard_normal((nI, d)) * 0.4
    users = rng.integers(0, nU, B)
    pos, neg = rng.integers(0, nI, B), rng.integers(0, nI, B)
    pp = rng.uniform(0.1, 1, B) if with_pop else None
    pn = rng.uniform(0.1, 1, B) if with_pop else None
    fw = po.bpr_forward(U, I, users, pos, neg, pp, pn)
    loss, mf, reg = po.bpr_loss(fw, regs, B)
    gU, gI = po.dense_grads(nU, nI, users, pos, neg, *po.bpr_grads(fw, regs, B, pp, pn))

    Ut, It = torch.tensor(U, requires_grad=True), torch.tensor(I, requires_grad=True)
    ue, pe, ne = Ut[users], It[pos], It[neg]
    ps, ns = (ue * pe).sum(1), (ue * ne).sum(1)
    if with_pop:
        ps = (torch.nn.functional.elu(ps) + 1) * torch.tensor(pp)
        ns = (torch.nn.functional.elu(ns) + 1) * torch.tensor(pn)
    mf_t = -torch.log(torch.sigmoid(ps - ns) + 1e-10).mean()
    reg_t = regs * 0.5 * ((ue ** 2).sum() + (pe ** 2).sum() + (ne ** 2).sum()) / B
    (mf_t + reg_t).backward()
    np.testing.assert_allclose([loss, mf, reg], [float((mf_t + reg_t).detach()), float(mf_t.detach()), float(reg_t.detach())], rtol=1e-12)
    np.testing.assert_allclose(gU, Ut.grad.numpy(), rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(gI, It.grad.numpy(), rtol=1e-9, atol=1e-14)


def test_adam_is_dense_decay_with_presummed_duplicates():
    """Independent scatter-style restatement of TF-1.14 `_apply_sparse_shared` [TF-ext] vs the oracle's dense formula."""
    rng = np.random.default_rng(6)
    n, d, lr = 20, 4, 1e-2
    var = rng.standard_normal((n, d))
    m = np.zeros((n, d))
    v = np.zeros((n, d))
    var2, m2, v2 = var.copy(), m.copy(), v.copy()
    b1, b2, eps = 0.9, 0.999, 1e-8
    for t in (1, 2, 3):
        idx = rng.integers(0, n, 12)                      # duplicates inside the batch
        vals = rng.standard_normal((12, d))
        dense = np.zeros((n, d))
        np.add.at(dense, idx, vals)
        var, m, v = po.adam_dense_decay_step(var, m, v, dense, t, lr)
        # scatter style: unsorted_segment_sum, decay everything, scatter_add, update everything
        uniq, inv = np.unique(idx, return_inverse=True)
        summed = np.zeros((len(uniq), d))
        np.add.at(summed, inv, vals)
        lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        m2 *= b1
        m2[uniq] += (1 - b1) * summed
        v2 *= b2
        v2[uniq] += (1 - b2) * summed * summed
        var2 -= lr_t * m2 / (np.sqrt(v2) + eps)
    np.testing.assert_allclose(var, var2, rtol=1e-13)
    np.testing.assert_allclose(m, m2, rtol=1e-13)
    np.testing.assert_allclose(v, v2, rtol=1e-13)


def test_build_eval_blocks_shapes():
    train = {0: [1, 2], 1: [], 2: [5, 5, 7], 3: [0]}
    blocks = po.build_eval_blocks({2: [9], 0: [3], 3: [4]}, train, block=2)
    assert [b[0] for b in blocks] == [[2, 0], [3]]
    np.testing.assert_array_equal(blocks[0][1], [[0, 5], [0, 5], [0, 7], [1, 1], [1, 2]])
    assert blocks[0][2:] == (2, 5) and blocks[1][2:] == (1, 1)
    with pytest.raises(KeyError):
        po.build_eval_blocks({9: [1]}, train)
