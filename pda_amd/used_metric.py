"""Ranking metrics with the reference's names (MF/used_metric.py).

`evaluate_on_device` is the product path: pda_metrics (HIP) over the whole top-K matrix, replacing the
reference's multiprocessing.Pool(5) that pickles every block (MF/train_new_api.py:741-778).
`get_performance` keeps the reference's per-user host signature for callers that still hold python lists
(MF/used_metric.py:69-80); it is host glue, not a fallback for the kernel.
"""
from __future__ import annotations

import numpy as np


def get_r(user_pos_test, r):
    return np.isin(np.asarray(r), np.asarray(list(user_pos_test))).astype(np.float64)


def get_performance(user_pos_test, r, Ks):
    hit = get_r(user_pos_test, r)
    n_pos = len(user_pos_test)
    out = {"recall": [], "precision": [], "ndcg": [], "hit_ratio": []}
    with np.errstate(divide="ignore", invalid="ignore"):
        for K in Ks:
            assert K >= 1
            h = hit[:K]
            disc = 1.0 / np.log2(np.arange(2, K + 2))
            ideal = disc[:min(n_pos, K)].sum()
            out["precision"].append(np.mean(h))
            out["recall"].append(np.sum(h) / n_pos)
            out["ndcg"].append(0.0 if not ideal else float((h * disc[:h.size]).sum() / ideal))
            out["hit_ratio"].append(min(1.0, np.sum(h)))
    return {k: np.array(v) for k, v in out.items()}


def evaluate_on_device(topk, tgt_indptr, tgt_indices, Ks, tot_user=None):
    """topk i32 [n,k] (cuda), targets CSR by row (cuda) -> dict of float64 numpy arrays [len(Ks)] = sums / tot_user."""
    import torch
    from . import ops
    ks = torch.as_tensor(list(Ks), dtype=torch.int32, device=topk.device)
    sums = ops.metrics_sums(topk, tgt_indptr, tgt_indices, ks).cpu().numpy()
    n = float(tot_user if tot_user is not None else topk.shape[0])
    return {"precision": sums[0] / n, "recall": sums[1] / n, "ndcg": sums[2] / n, "hit_ratio": sums[3] / n}
