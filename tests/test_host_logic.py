"""CPU suite, part 3: host-side logic of the drop-in (no kernels): sharding arithmetic, samplers, dataset writer +
loaders round trip, early stopping, the Session/fetch protocol."""
import os
import random

import numpy as np
import pytest

from pda_amd import dist, load_data, parse, sampler, synthetic


def test_shard_ranges_cover_and_balance():
    for n in (1, 31, 32, 33, 20000, 200000, 2_000_001):
        for w in (1, 2, 3, 4, 8):
            rs = [dist.shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            sizes = [hi - lo for lo, hi in rs]
            assert all(lo % 32 == 0 for lo, hi in rs if hi > lo)            # non-empty shards start on a 32-item tile
            full = [s for s in sizes if s > 0][:-1]                           # every non-empty shard but the last is "per" items
            assert len(set(full)) <= 1 and all(s % 32 == 0 for s in full)


@pytest.fixture(scope="module")
def toy(tmp_path_factory):
    root = tmp_path_factory.mktemp("data")
    synthetic.write_dataset(str(root / "toy"), n_users=120, n_items=90, mean_hist=12)
    return str(root) + "/"


def test_written_dataset_round_trips_through_both_loaders(toy):
    a = parse.parse_args(["--data_path", toy, "--dataset", "toy", "--batch_size", "32", "--train", "s_condition"])
    d2, d1 = load_data.Data2(a), load_data.Data(a)
    assert d1.n_train == d2.n_train and d1.n_users == d2.n_users == 120 and d1.n_items == d2.n_items == 90
    for u, items in d1.train_user_list.items():
        assert sorted(items) == sorted(d2.train_user_list[u])
        assert len(d2.train_user_list_time[u]) == len(items)
    assert len(d2.unique_times) == 9 and set(d2.unique_times) == set(range(9))
    pop = load_data.load_popularity(a)
    assert pop.shape == (90, 10) and pop.min() == 0.0 and pop.max() == 1.0
    ip, ix, sl = d2.train_csr("cpu")
    assert int(ip[-1]) == d2.n_train
    for u in (0, 7, 119):
        row = ix[ip[u]:ip[u + 1]].numpy()
        assert np.all(np.diff(row) >= 0) and sorted(d2.train_user_list[u]) == row.tolist()
        # slots stay attached to their items through the sort
        pairs = sorted(zip(d2.train_user_list[u], d2.train_user_list_time[u]))
        assert sorted(zip(row.tolist(), sl[ip[u]:ip[u + 1]].tolist())) == pairs


def test_host_generator_follows_the_sampler_protocol(toy):
    a = parse.parse_args(["--data_path", toy, "--dataset", "toy", "--batch_size", "32", "--train", "s_condition"])
    d = load_data.Data2(a)
    pop = np.power(load_data.get_popularity_from_load(load_data.load_popularity(a)), 0.22)
    d.add_expo_popularity(pop)
    random.seed(2020)
    np.random.seed(2020)
    batches = list(sampler.host_generator(d, with_pop=True))
    assert len(batches) == d.n_train // 32 + 1                       # MF/train_new_api.py:190
    for users, pos, neg, pp, pn in batches[:5]:
        assert len(users) == len(pos) == len(neg) == len(pp) == len(pn) == 32
        assert len(set(users)) == 32                                 # rd.sample: unique users
        for u, p, n, a_, b_ in zip(users, pos, neg, pp, pn):
            assert p in d.train_user_list[u] and n not in d.train_user_list[u]
            ts = [t for i, t in zip(d.train_user_list[u], d.train_user_list_time[u]) if i == p]
            assert any(a_ == pop[p, t] and b_ == pop[n, t] for t in ts)
    plain = next(iter(sampler.host_generator(d, with_pop=False)))
    assert len(plain) == 3


def test_early_stop_matches_reference_logic():
    from pda_amd.train_new_api import early_stop
    cfg = dict(best_hr=0, best_ndcg=0, best_recall=0, best_pre=0, best_epoch=0)
    step = 0
    cfg, step, stop = early_stop(0.3, 0.2, 0.1, 0.05, 0, cfg, step, flag_step=2)
    assert (step, stop, cfg["best_recall"], cfg["best_epoch"]) == (0, False, 0.1, 0)
    cfg, step, stop = early_stop(0.3, 0.2, 0.1, 0.05, 5, cfg, step, flag_step=2)     # ties count as improvement (>=)
    assert (step, cfg["best_epoch"]) == (0, 5)
    cfg, step, stop = early_stop(0.1, 0.1, 0.05, 0.01, 10, cfg, step, flag_step=2)
    assert (step, stop) == (1, False)
    cfg, step, stop = early_stop(0.1, 0.1, 0.05, 0.01, 15, cfg, step, flag_step=2)
    assert (step, stop, cfg["best_epoch"]) == (2, True, 5)


def test_unknown_modes_raise_like_the_reference():
    from pda_amd import train_new_api as t
    a = parse.parse_args(["--train", "temp_pop"])
    with pytest.raises(NotImplementedError):
        t.DatasetApi_Model(a, {"n_users": 4, "n_items": 4}, 16, lambda: iter(()), device="cpu")


def test_sweep_kernel_and_geometry_selection(monkeypatch):
    """Which kernel generation and which workgroup geometry a score call gets (pda_amd.ops.score_kernel / few_candidates_hint):
    pure host logic, results never depend on it -- but the measured defaults should not drift unnoticed."""
    from pda_amd import ops
    for var in ("PDA_SCORE_KERNEL", "PDA_SCORE_LISTS", "PDA_SCORE_PRUNE"):
        monkeypatch.delenv(var, raising=False)
    P, R = ops.HEAD_POP, ops.HEAD_RAW
    # generation: v4 wherever it fits, except natural order / raw head at d = 256
    assert ops.score_kernel(128, 50, 200000, True, P) == "v4"
    assert ops.score_kernel(128, 50, 200000, "order", P) == "v4"
    assert ops.score_kernel(128, 50, 200000, False, P) == "v4"          # natural order: many-candidates geometry (round 3)
    assert ops.score_kernel(64, 50, 20000, "order", R) == "v4"
    assert ops.score_kernel(256, 50, 250000, True, P) == "v4"
    assert ops.score_kernel(256, 50, 250000, False, P) == "v3"
    assert ops.score_kernel(256, 50, 250000, "order", R) == "v3"
    assert ops.score_kernel(32, 50, 1000, True, P) == "v3"
    # geometry hints (bits of early_stop): 128 = huge, 8 = many candidates, 0 = default
    assert ops.few_candidates_hint(P, "order", 262144, 128) == 128        # the headline: dense sweep of a very large block (1 024-user workgroups)
    assert ops.few_candidates_hint(P, "order", 196608, 128) == 0          # (a caller's own split count, fewer than 192 x 1 024 + 1 users: the default geometry;
    assert ops.few_candidates_hint(P, "order", 65536, 128) == 0           #  rounds 3 / 4 had a wide geometry here)
    assert ops.few_candidates_hint(P, "order", 262144, 256) == 0
    # ... and with the split count the library itself picks (score_topk_keys with n_splits left to it): the huge geometry from 4 096 users on,
    # item splits by rounds of 256 workgroups (one shared warm-up: splits are cheap)
    assert [ops.huge_splits(u, 200000) for u in (2048, 8192, 32768, 65536, 98304, 163840, 196608, 262144)] == [0, 32, 8, 4, 8, 3, 4, 1]
    assert ops.huge_splits(50000, 20000) == 5 and ops.huge_splits(47890, 26047) == 5 and ops.huge_splits(262144, 25000) == 1
    assert ops.few_candidates_hint(P, "order", 65536, 128, 200000, 4) == 128 and ops.few_candidates_hint(P, "order", 65536, 128, 200000, 2) == 0
    assert ops.few_candidates_hint(P, "order", 50000, 64, 20000, 5) == 128
    # d = 256 (config 5): 512-user workgroups
    assert ops.huge_splits(262144, 250000, 256) == 1 and ops.huge_splits(65536, 250000, 256) == 2
    assert ops.few_candidates_hint(P, "order", 262144, 256, 250000, 1) == 128 and ops.few_candidates_hint(P, True, 262144, 256, 250000, 1) == 0
    assert ops.few_candidates_hint(P, True, 262144, 128) == 0             # early-terminating: a warm-up and a sort
    assert ops.few_candidates_hint(P, False, 50000, 64) == 8              # natural order
    assert ops.few_candidates_hint(R, "order", 262144, 128) == 8          # raw head by norm
    assert ops.few_candidates_hint(R, False, 65536, 128) == 8
    assert ops.few_candidates_hint(R, True, 65536, 128) == 0
    assert ops.few_candidates_hint(R, "order", 65536, 256) == 0
    assert ops.prune_default(P, 128) is True and ops.prune_default(R, 128) == "order" and ops.prune_default(R, 256) is False
    monkeypatch.setenv("PDA_SCORE_LISTS", "many")
    assert ops.few_candidates_hint(P, "order", 262144, 128) == 8


def test_the_library_plans_what_the_python_rules_chose(monkeypatch):
    """pda_score_topk_plan (the policy behind the C ABI since round 5) against the Python rules of rounds 2 - 4 on their table of cases: kernel
    generation, sweep mode, item splits, geometry hint; and the raw head on large blocks goes to the funnel."""
    from pda_amd import ops
    for k in ("PDA_SCORE_KERNEL", "PDA_SCORE_LISTS", "PDA_SCORE_PRUNE", "PDA_HUGE_SPLITS", "PDA_SCORE_FUNNEL"):
        monkeypatch.delenv(k, raising=False)
    P, R = ops.HEAD_POP, ops.HEAD_RAW
    lib = ops._lib.load()
    cases = [(nu, nI, d, head, prune) for (nu, nI, d) in ((262144, 200000, 128), (196608, 200000, 128), (98304, 200000, 128), (65536, 200000, 128),
                                                           (20000, 200000, 128), (2048, 200000, 128), (50000, 20000, 64), (47890, 26047, 64),
                                                           (262144, 250000, 256), (8192, 250000, 256), (1000, 3000, 32), (262144, 25000, 128))
             for head in (P, R) for prune in (None, False, True, "order")]
    n_funnel = 0
    for nu, nI, d, head, prune in cases:
        plan = ops.score_plan(nu, nI, d, 50, head, prune)
        pr = ops.prune_default(head, d) if prune is None else prune
        assert plan["sweep_mode"] == pr, (nu, nI, d, head, prune, plan)
        if ops.score_impl(d, 50, nI) == "v1":
            assert plan["kernel"] == "v1"
            continue
        if ops.funnel_applies(d, 50, nu, nI, head, pr, None):
            assert plan["kernel"] == "funnel" and plan["n_splits"] == 1 and plan["order"] == 3 and plan["workspace_bytes"] == lib.pda_score_topk7_workspace_bytes(nu, nI, d)
            n_funnel += 1
            continue
        gen = ops.score_kernel(d, 50, nI, pr, head)
        assert plan["kernel"] == gen, (nu, nI, d, head, prune, plan)
        if gen != "v4":
            continue
        ns = lib.pda_score_topk4_auto_splits(nu, nI, d)
        if head == P and pr == "order":
            ns = ops.huge_splits(nu, nI, d) or ns
        es = (1 if pr is True else 0) | ops.few_candidates_hint(head, pr, nu, d, nI, ns)
        assert (plan["n_splits"], plan["early_stop"]) == (ns, es), (nu, nI, d, head, prune, plan, ns, es)
        assert plan["workspace_bytes"] == lib.pda_score_topk4_workspace_bytes(nu, nI, d, ns) and plan["keys_rows"] == ns * nu
        assert plan["order"] == (0 if pr is False else (1 if head == P else 2)) and plan["prep_with_pop"] == (head == P)
    assert n_funnel >= 6                     # (262 144 / 196 608 / 98 304 / 65 536 users x 200 000 items, raw head, default / natural / dense modes)
    # the huge geometry's item splits: the library's rule IS the old Python rule
    assert [ops.huge_splits(u, 200000) for u in (2048, 8192, 32768, 65536, 98304, 163840, 196608, 262144)] == [0, 32, 8, 4, 8, 3, 4, 1]
