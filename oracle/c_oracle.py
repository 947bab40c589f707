"""ctypes front-end for oracle/libpda_oracle.so and oracle/_ref/libpda_ref.so.

TEST INFRASTRUCTURE ONLY (see oracle/pda_oracle.py header).  Build with ``make -C oracle``
(``__graft_entry__.build()`` does it).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


def _ptr(a, ctype):
    return None if a is None else a.ctypes.data_as(C.POINTER(ctype))


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libpda_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.oracle_score_topk.restype = C.c_int
    return _LIB


def ref_lib():
    """The reference's own arg_topk.h / evaluate.h compiled by oracle/Makefile (None if absent)."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libpda_ref.so")
        if not os.path.exists(path):
            return None
        _REF = C.CDLL(path)
    return _REF


def score_topk(U, I, users, K=50, mode=0, pop=None, hist_indptr=None, hist_indices=None,
               item_offset=0, n_items_local=None, order=0, want_scores=False):
    """oracle_score_topk: returns (idx int32[B,K], val float32[B,K][, scores float32[B,n_local]])."""
    U = np.ascontiguousarray(U, dtype=np.float32)
    I = np.ascontiguousarray(I, dtype=np.float32)
    users = np.ascontiguousarray(users, dtype=np.int32)
    n_local = I.shape[0] - item_offset if n_items_local is None else n_items_local
    if pop is not None:
        pop = np.ascontiguousarray(pop, dtype=np.float32)
    if hist_indptr is not None:
        hist_indptr = np.ascontiguousarray(hist_indptr, dtype=np.int64)
        hist_indices = np.ascontiguousarray(hist_indices, dtype=np.int32)
    B = users.shape[0]
    idx = np.empty((B, K), dtype=np.int32)
    val = np.empty((B, K), dtype=np.float32)
    sc = np.empty((B, n_local), dtype=np.float32) if want_scores else None
    rc = lib().oracle_score_topk(_ptr(U, C.c_float), _ptr(I, C.c_float), _ptr(pop, C.c_float),
                                 _ptr(users, C.c_int32), C.c_int(B), C.c_int(n_local), C.c_int(item_offset),
                                 C.c_int(U.shape[1]), _ptr(hist_indptr, C.c_int64), _ptr(hist_indices, C.c_int32),
                                 C.c_int(K), C.c_int(mode), C.c_int(order), _ptr(idx, C.c_int32),
                                 _ptr(val, C.c_float), _ptr(sc, C.c_float))
    if rc != 0:
        raise ValueError("oracle_score_topk: bad arguments (K>n_items, K<1 or d%8)")
    return (idx, val, sc) if want_scores else (idx, val)


def cpu_port_score_topk(U, I, users, K=50, mode=0, pop=None, hist_indptr=None, hist_indices=None):
    """cpu_port_score_topk (oracle/pda_cpu_port.c): the native fused CPU baseline, all OpenMP threads.  U, I float32 C-contiguous
    (numpy or the .numpy() view of a torch CPU tensor: no copy); hist CSR by BLOCK row.  -> (idx int32 [B, K], val float32 [B, K])."""
    U = np.ascontiguousarray(U, dtype=np.float32)
    I = np.ascontiguousarray(I, dtype=np.float32)
    users = np.ascontiguousarray(users, dtype=np.int32)
    if pop is not None:
        pop = np.ascontiguousarray(pop, dtype=np.float32)
    if hist_indptr is not None:
        hist_indptr = np.ascontiguousarray(hist_indptr, dtype=np.int64)
        hist_indices = np.ascontiguousarray(hist_indices, dtype=np.int32)
    B = users.shape[0]
    idx = np.empty((B, K), dtype=np.int32)
    val = np.empty((B, K), dtype=np.float32)
    rc = lib().cpu_port_score_topk(_ptr(U, C.c_float), _ptr(I, C.c_float), _ptr(pop, C.c_float), _ptr(users, C.c_int32), C.c_int(B),
                                   C.c_int(I.shape[0]), C.c_int(U.shape[1]), _ptr(hist_indptr, C.c_int64), _ptr(hist_indices, C.c_int32),
                                   C.c_int(K), C.c_int(mode), _ptr(idx, C.c_int32), _ptr(val, C.c_float))
    if rc != 0:
        raise ValueError("cpu_port_score_topk: bad arguments")
    return idx, val


def arg_topk_2d(ratings, K=50):
    """oracle_arg_topk_2d: the native CPU top-K baseline (all OpenMP threads).  ratings float32 [rows, n] -> int32 [rows, K]."""
    ratings = np.ascontiguousarray(ratings, dtype=np.float32)
    rows, n = ratings.shape
    out = np.empty((rows, K), dtype=np.int32)
    lib().oracle_arg_topk_2d(_ptr(ratings, C.c_float), C.c_int(n), C.c_int(rows), C.c_int(K), _ptr(out, C.c_int32))
    return out


def scores_chain(U, I, users):
    U = np.ascontiguousarray(U, dtype=np.float32)
    I = np.ascontiguousarray(I, dtype=np.float32)
    users = np.ascontiguousarray(users, dtype=np.int32)
    out = np.empty((users.shape[0], I.shape[0]), dtype=np.float32)
    lib().oracle_scores_chain(_ptr(U, C.c_float), _ptr(I, C.c_float), _ptr(users, C.c_int32),
                              C.c_int(users.shape[0]), C.c_int(I.shape[0]), C.c_int(U.shape[1]), _ptr(out, C.c_float))
    return out


def metrics(topk, tgt_indptr, tgt_indices, Ks):
    """Sums (not yet divided by tot_user) of precision, recall, ndcg, hit: float64 [4, len(Ks)]."""
    topk = np.ascontiguousarray(topk, dtype=np.int32)
    tgt_indptr = np.ascontiguousarray(tgt_indptr, dtype=np.int64)
    tgt_indices = np.ascontiguousarray(tgt_indices, dtype=np.int32)
    Ks = np.ascontiguousarray(Ks, dtype=np.int32)
    sums = np.zeros((4, Ks.shape[0]), dtype=np.float64)
    lib().oracle_metrics(_ptr(topk, C.c_int32), C.c_int(topk.shape[0]), C.c_int(topk.shape[1]),
                         _ptr(tgt_indptr, C.c_int64), _ptr(tgt_indices, C.c_int32), _ptr(Ks, C.c_int32),
                         C.c_int(Ks.shape[0]), _ptr(sums, C.c_double))
    return sums


def ref_arg_topk(ratings, top_k, threads=4):
    """Reference arg_top_k_2d (util/cython/include/arg_topk.h:29).  None if oracle/_ref is absent."""
    r = ref_lib()
    if r is None:
        return None
    ratings = np.ascontiguousarray(ratings, dtype=np.float32)
    out = np.empty((ratings.shape[0], top_k), dtype=np.int32)
    r.ref_arg_top_k_2d(_ptr(ratings, C.c_float), C.c_int(ratings.shape[1]), C.c_int(ratings.shape[0]),
                       C.c_int(top_k), C.c_int(threads), _ptr(out, C.c_int))
    return out


def ref_evaluate_matrix(ratings, tgt_indptr, tgt_indices, metric_ids, top_k, threads=4):
    """Reference cpp_evaluate_matrix (evaluator/backend/cpp/include/evaluate.h:53).  [n_users, n_metric, top_k]."""
    r = ref_lib()
    if r is None:
        return None
    ratings = np.ascontiguousarray(ratings, dtype=np.float32)
    tgt_indptr = np.ascontiguousarray(tgt_indptr, dtype=np.int64)
    tgt_indices = np.ascontiguousarray(tgt_indices, dtype=np.int32)
    m = np.ascontiguousarray(metric_ids, dtype=np.int32)
    out = np.zeros((ratings.shape[0], m.shape[0], top_k), dtype=np.float32)
    r.ref_cpp_evaluate_matrix(_ptr(ratings, C.c_float), C.c_int(ratings.shape[1]), C.c_int(ratings.shape[0]),
                              _ptr(tgt_indptr, C.c_longlong), _ptr(tgt_indices, C.c_int), _ptr(m, C.c_int),
                              C.c_int(m.shape[0]), C.c_int(top_k), C.c_int(threads), _ptr(out, C.c_float))
    return out
