#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by RUNNING the reference's own runnable pieces.

Run in the authoring container only (needs /root/reference and oracle/_ref):  python tests/golden/make_golden.py
Nothing of the reference is copied: the fixtures hold inputs we invent here and the outputs the reference
code produced for them.

  metrics.json     MF/used_metric.py:get_performance (np.float shim for numpy>=1.24) on random + edge cases,
                   cross-checked against evaluator/backend/python/metric.py at the sampled K
  pop_pre.json     pop_pre.py run as a script on toy t_k.txt files -> item_pop_seq_ori2.txt contents
  loader.json      MF/load_data.py:Data on toy train/valid/test.txt (attributes the trainer reads)
  topk_ref.npz     util/cython/include/arg_topk.h:arg_top_k_2d and evaluator/backend/cpp evaluate.h:
                   cpp_evaluate_matrix (compiled into oracle/_ref) on tie-free score matrices
  flags.json       argparse namespace of MF/parse.py with no arguments (flag names and defaults)
  popularity_heads.npz   NOT a reference run: the numpy expressions of MF/train_new_api.py:954-959,988-990 TRANSCRIBED by the builder
                         (that file imports TensorFlow and cannot be imported here).  It pins our host code against a second writing
                         of the same three lines, nothing more; every other fixture above is an output of the reference's own code.
"""
import importlib.util
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def golden_metrics():
    np.float = float                                  # numpy>=1.24 removed the alias used at used_metric.py:66
    um = load_by_path("ref_used_metric", os.path.join(REF, "MF/used_metric.py"))
    pm = load_by_path("ref_py_metric", os.path.join(REF, "evaluator/backend/python/metric.py"))
    rng = np.random.default_rng(7)
    cases = []
    specs = [([3, 5], [5, 1, 2, 3], [2, 4])]          # the survey's probe
    for _ in range(40):
        n_items = int(rng.integers(60, 400))
        r = rng.permutation(n_items)[:50].tolist()
        n_t = int(rng.integers(1, 70))
        tgt = rng.permutation(n_items)[:n_t].tolist()
        if rng.random() < 0.5:
            tgt[: min(4, n_t)] = [r[i] for i in rng.permutation(50)[: min(4, n_t)]]
        specs.append((tgt, r, [20, 50] if rng.random() < 0.7 else [1, 5, 10, 50]))
    specs.append(([999], list(range(50)), [20, 50]))            # zero hits
    specs.append((list(range(50)), list(range(50)), [20, 50]))  # all hits, len(target) == K
    specs.append((list(range(3)), list(range(50)), [20, 50]))   # len(target) < K
    for tgt, r, Ks in specs:
        out = um.get_performance(tgt, r, Ks)
        if len(r) == 50 and max(Ks) <= 50:                      # second reference implementation agrees at K
            hit_rank = pm.ndcg(r, tgt)
            for q, K in enumerate(Ks):
                assert abs(hit_rank[K - 1] - out["ndcg"][q]) < 1e-9, (K, hit_rank[K - 1], out["ndcg"][q])
                assert abs(pm.recall(r, tgt)[K - 1] - out["recall"][q]) < 1e-6      # float32 cumsum there
                assert abs(pm.precision(r, tgt)[K - 1] - out["precision"][q]) < 1e-6
        cases.append({"target": tgt, "r": r, "Ks": Ks, "out": {k: v.tolist() for k, v in out.items()}})
    json.dump(cases, open(os.path.join(HERE, "metrics.json"), "w"))
    print("metrics.json:", len(cases), "cases")


def golden_pop_pre():
    rng = np.random.default_rng(11)
    n_item, T = 23, 5
    files = {}
    with tempfile.TemporaryDirectory() as d:
        for t in range(T):
            lines = []
            present = rng.permutation(n_item)[: rng.integers(8, n_item + 1)]
            if t == 0:
                present = np.arange(n_item)                      # every id occurs at least once
            for it in present:
                users = rng.integers(0, 500, rng.integers(1, 30)).tolist()
                lines.append(" ".join(str(x) for x in [int(it)] + users))
            files["t_%d.txt" % t] = "\n".join(lines) + "\n"
            open(os.path.join(d, "t_%d.txt" % t), "w").write(files["t_%d.txt" % t])
        subprocess.check_call([sys.executable, os.path.join(REF, "pop_pre.py"), "--path", d + "/", "--slot_count", str(T)],
                              stdout=subprocess.DEVNULL)
        out = open(os.path.join(d, "item_pop_seq_ori2.txt")).read()
    json.dump({"slot_count": T, "files": files, "item_pop_seq_ori2": out}, open(os.path.join(HERE, "pop_pre.json"), "w"))
    print("pop_pre.json:", len(out.splitlines()), "items")


def golden_loader():
    rng = np.random.default_rng(13)
    files = {}
    for name, n_lines in (("train.txt", 30), ("valid.txt", 9), ("test.txt", 14)):
        lines = []
        for u in rng.permutation(40)[:n_lines]:
            items = rng.integers(0, 57, rng.integers(1, 12)).tolist()
            lines.append(" ".join(str(x) for x in [int(u)] + items))
        files[name] = "\n".join(lines) + "\n"
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "data", "toy"))
        for k, v in files.items():
            open(os.path.join(d, "data", "toy", k), "w").write(v)
        code = ("import sys,json; sys.path.insert(0,'%s/MF'); sys.argv=['x','--dataset','toy','--batch_size','8'];"
                "from parse import parse_args; from load_data import Data; a=parse_args(); D=Data(a);"
                "print('JSON'+json.dumps(dict(n_users=D.n_users,n_items=D.n_items,n_train=D.n_train,n_valid=D.n_valid,"
                "n_test=D.n_test,train={str(k):v for k,v in D.train_user_list.items()},"
                "valid={str(k):v for k,v in D.valid_user_list.items()},test={str(k):v for k,v in D.test_user_list.items()},"
                "n_items_list=len(D.items),batch_size=D.batch_size)))") % REF
        out = subprocess.check_output([sys.executable, "-c", code], cwd=d).decode()
    exp = json.loads([l for l in out.splitlines() if l.startswith("JSON")][0][4:])
    json.dump({"files": files, "expected": exp}, open(os.path.join(HERE, "loader.json"), "w"))
    print("loader.json: n_users", exp["n_users"], "n_items", exp["n_items"], "n_train", exp["n_train"])


def golden_topk_ref():
    from oracle import c_oracle
    assert c_oracle.ref_lib() is not None, "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(17)
    scores = rng.standard_normal((37, 211)).astype(np.float32)          # continuous => tie-free
    top = c_oracle.ref_arg_topk(scores, 50)
    lens = rng.integers(1, 30, 37)
    indptr = np.zeros(38, np.int64)
    indptr[1:] = np.cumsum(lens)
    tgt = np.concatenate([rng.permutation(211)[:n] for n in lens]).astype(np.int32)
    res = c_oracle.ref_evaluate_matrix(scores, indptr, tgt, [1, 2, 4], 50)   # precision, recall, ndcg (cumulative per rank)
    np.savez_compressed(os.path.join(HERE, "topk_ref.npz"), scores=scores, arg_topk=top, tgt_indptr=indptr, tgt_indices=tgt,
                        eval_matrix=res)
    print("topk_ref.npz:", top.shape, res.shape)


def golden_popularity_heads():
    rng = np.random.default_rng(19)
    pop_item_all = rng.uniform(0, 1, (31, 10))
    pop_item_all[rng.integers(0, 31, 4), rng.integers(0, 10, 4)] = 0.0
    pop_item_all[3, -2], pop_item_all[3, -3] = 0.9, 0.1                   # linear prediction > 1 -> clipped
    pop_item_all[4, -2], pop_item_all[4, -3] = 0.05, 0.9                  # linear prediction <= 0 -> 1e-9
    g = 0.22
    # literal transcription of the expressions' *values* (MF/train_new_api.py:954-959, 988-990)
    last = np.power(pop_item_all[:, -2], g)
    lin = pop_item_all[:, -2] + 0.5 * (pop_item_all[:, -2] - pop_item_all[:, -3])
    lin[np.where(lin <= 0)] = 1e-9
    lin[np.where(lin > 1.0)] = 1.0
    lin = np.power(lin, g)
    train = np.power(pop_item_all[:, :-1], g)
    np.savez_compressed(os.path.join(HERE, "popularity_heads.npz"), pop_item_all=pop_item_all, gamma=g, last=last, linear=lin,
                        train=train)
    print("popularity_heads.npz ok")


def golden_flags():
    """Flag names + defaults of MF/parse.py (argparse only, imports cleanly)."""
    old = sys.argv
    sys.argv = ["x"]
    try:
        ref_parse = load_by_path("ref_parse", os.path.join(REF, "MF/parse.py"))
        ns = vars(ref_parse.parse_args())
    finally:
        sys.argv = old
    json.dump(ns, open(os.path.join(HERE, "flags.json"), "w"), sort_keys=True)
    print("flags.json:", len(ns), "flags")


if __name__ == "__main__":
    golden_flags()
    golden_metrics()
    golden_pop_pre()
    golden_loader()
    golden_topk_ref()
    golden_popularity_heads()
