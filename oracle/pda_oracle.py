"""CPU oracle for the PDA BPR-MF hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain numpy restatement of the arithmetic the reference expresses as a
TensorFlow-1.14 graph.  It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under ``pda_amd/``
imports it, and the product path raises if the HIP library is missing.

PARITY PINNING STATUS
---------------------
* The TF graph itself (forward / loss / grads / Adam / top-k) cannot be executed here or on
  the GPU box (TensorFlow 1.14 is absent, un-vendored, un-installable).  The reference has
  no tests or golden vectors for it.  Those functions are therefore **parity unpinned**
  against the reference; they are pinned instead by an independent double oracle:
  closed-form float64 numpy (this file)  <->  torch.autograd float32/float64
  (tests/test_oracle.py) and, for top-k on tie-free rows, by the reference's own
  C++ ``arg_top_k_2d`` compiled into ``oracle/_ref`` (tests/test_oracle_ref.py).
* Metrics, popularity pre-compute and the text loaders ARE pinned: against golden
  vectors generated in the authoring container by importing the reference's runnable
  Python (``tests/golden/make_golden.py`` -> ``tests/golden/*.json|npz``).

Every function cites the reference lines it follows (paths relative to the reference
repo root).  [TF-ext] marks semantics that live in TensorFlow 1.14, not in the tree.
"""
from __future__ import annotations

import numpy as np

ADAM_BETA1 = 0.9      # tf.train.AdamOptimizer defaults [TF-ext]
ADAM_BETA2 = 0.999
ADAM_EPS = 1e-8


# --------------------------------------------------------------------------------------
# A1/A2: gather + forward                       MF/model_api.py:51-53, 102-110, 123-125
# --------------------------------------------------------------------------------------
def elu_plus_one(x):
    """tf.nn.elu(x) + 1  ==  x+1 if x>0 else exp(x)   [TF-ext]; MF/model_api.py:107-108,113."""
    x = np.asarray(x)
    return np.where(x > 0, x + 1.0, np.exp(np.minimum(x, 0.0)))


def bpr_forward(U, I, users, pos, neg, pos_pop=None, neg_pop=None, dtype=np.float64):
    """Gather three rows per triplet, two dots, optional (ELU+1)*pop re-weighting.

    ``pos_pop is None``  -> plain BPRMF   (MF/model_api.py:695-697, identical copy :123-125)
    otherwise            -> PD/PDA        (MF/model_api.py:102-110)
    Returns dict(ue, pe, ne, ps, ns, psw, nsw) with psw/nsw the scores fed to the loss.
    """
    U = np.asarray(U, dtype=dtype)
    I = np.asarray(I, dtype=dtype)
    ue, pe, ne = U[users], I[pos], I[neg]            # tf.nn.embedding_lookup, :51-53
    ps = (ue * pe).sum(axis=1)                       # tf.reduce_sum(tf.multiply(..)), :103
    ns = (ue * ne).sum(axis=1)                       # :104
    if pos_pop is None:
        psw, nsw = ps, ns
    else:
        psw = elu_plus_one(ps) * np.asarray(pos_pop, dtype=dtype)   # :107,109
        nsw = elu_plus_one(ns) * np.asarray(neg_pop, dtype=dtype)   # :108,110
    return dict(ue=ue, pe=pe, ne=ne, ps=ps, ns=ns, psw=psw, nsw=nsw)


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# --------------------------------------------------------------------------------------
# A3: loss                                        MF/model_api.py:112-121 / :699-705
# --------------------------------------------------------------------------------------
def bpr_loss(fw, regs, batch_size):
    """mf = -mean(log(sigmoid(psw-nsw)+1e-10));  reg = regs*(l2(ue)+l2(pe)+l2(ne))/batch_size

    tf.nn.l2_loss(x) = sum(x**2)/2 [TF-ext].  ``batch_size`` is the *flag* constant
    (MF/model_api.py:118), not the runtime batch length.
    Returns (loss, mf_loss, reg_loss) as python floats (float64 arithmetic).
    """
    x = fw["psw"] - fw["nsw"]
    maxi = np.log(_sigmoid(x) + 1e-10)                                   # :112 / :702
    mf = -np.mean(maxi)                                                  # :114 / :704
    l2 = 0.5 * ((fw["ue"] ** 2).sum() + (fw["pe"] ** 2).sum() + (fw["ne"] ** 2).sum())  # :117
    reg = regs * l2 / batch_size                                         # :118-120
    return float(mf + reg), float(mf), float(reg)


# --------------------------------------------------------------------------------------
# A4: closed-form gradient of A2-A3 (what tf `minimize` differentiates; :83, :471)
# --------------------------------------------------------------------------------------
def bpr_grads(fw, regs, batch_size, pos_pop=None, neg_pop=None):
    """Per-occurrence gradients wrt the three gathered rows.

    g   = d mf / d x            = -(1/B) * s(1-s)/(s+1e-10),  s = sigmoid(x), B = len(batch)
    a_p = pos_pop * (1 if ps>0 else exp(ps))   (ELU grad [TF-ext]);  a_p = 1 for BPRMF
    d ue = g*(a_p*pe - a_n*ne) + (regs/batch_size)*ue
    d pe = g*a_p*ue            + (regs/batch_size)*pe
    d ne = -g*a_n*ue           + (regs/batch_size)*ne
    Note the mean is over the runtime batch (tf.reduce_mean) while the reg divisor is the
    flag constant (MF/model_api.py:118); the sampler only emits full batches so both are B.
    """
    n = fw["ps"].shape[0]
    x = fw["psw"] - fw["nsw"]
    s = _sigmoid(x)
    g = -(1.0 / n) * s * (1.0 - s) / (s + 1e-10)
    if pos_pop is None:
        a_p = np.ones_like(g)
        a_n = np.ones_like(g)
    else:
        a_p = np.asarray(pos_pop, dtype=g.dtype) * np.where(fw["ps"] > 0, 1.0, np.exp(np.minimum(fw["ps"], 0.0)))
        a_n = np.asarray(neg_pop, dtype=g.dtype) * np.where(fw["ns"] > 0, 1.0, np.exp(np.minimum(fw["ns"], 0.0)))
    c = regs / batch_size
    due = (g * a_p)[:, None] * fw["pe"] - (g * a_n)[:, None] * fw["ne"] + c * fw["ue"]
    dpe = (g * a_p)[:, None] * fw["ue"] + c * fw["pe"]
    dne = -(g * a_n)[:, None] * fw["ue"] + c * fw["ne"]
    return due, dpe, dne


def dense_grads(n_users, n_items, users, pos, neg, due, dpe, dne):
    """Sum per-occurrence slices into dense [U,d]/[I,d] gradients.

    TF concatenates the pos and neg IndexedSlices on item_embedding and de-duplicates with
    unsorted_segment_sum before the optimiser [TF-ext]; np.add.at is that sum.
    """
    d = due.shape[1]
    gU = np.zeros((n_users, d), dtype=due.dtype)
    gI = np.zeros((n_items, d), dtype=due.dtype)
    np.add.at(gU, users, due)
    np.add.at(gI, pos, dpe)
    np.add.at(gI, neg, dne)
    return gU, gI


# --------------------------------------------------------------------------------------
# A5: optimiser.  Reference = tf.train.AdamOptimizer(lr)  MF/model_api.py:83, :470-471
# --------------------------------------------------------------------------------------
def adam_dense_decay_step(var, m, v, grad_dense, t, lr, beta1=ADAM_BETA1, beta2=ADAM_BETA2, eps=ADAM_EPS):
    """TF-1.14 Adam `_apply_sparse_shared` semantics [TF-ext]: NOT lazy.

    m <- beta1*m (all rows); m[idx] += (1-beta1)*g ; v likewise with g^2 ;
    var <- var - lr_t * m/(sqrt(v)+eps)  for **every** row, with
    lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t), t = 1 for the first step.
    ``grad_dense`` is zero on untouched rows, so the dense formula below is identical.
    Returns new (var, m, v); dtype follows the inputs.
    """
    lr_t = lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    m = beta1 * m + (1.0 - beta1) * grad_dense
    v = beta2 * v + (1.0 - beta2) * grad_dense * grad_dense
    var = var - lr_t * m / (np.sqrt(v) + eps)
    return var, m, v


def sgd_step(var, grad_dense, lr):
    """Plain SGD (the north_star's fused scatter-update mode; a declared deviation from :83)."""
    return var - lr * grad_dense


def train_step(U, I, users, pos, neg, pos_pop, neg_pop, regs, batch_size, lr, optimizer="adam",
               state=None, t=1, dtype=np.float64):
    """One whole reference step: forward, loss, grads, optimiser.  MF/train_new_api.py:1078-1096.

    ``state`` = dict(mU, vU, mI, vI) for adam (created as zeros when None).
    Returns (U1, I1, state, (loss, mf, reg)).
    """
    U = np.asarray(U, dtype=dtype)
    I = np.asarray(I, dtype=dtype)
    fw = bpr_forward(U, I, users, pos, neg, pos_pop, neg_pop, dtype=dtype)
    losses = bpr_loss(fw, regs, batch_size)
    due, dpe, dne = bpr_grads(fw, regs, batch_size, pos_pop, neg_pop)
    gU, gI = dense_grads(U.shape[0], I.shape[0], users, pos, neg, due, dpe, dne)
    if optimizer == "sgd":
        return sgd_step(U, gU, lr), sgd_step(I, gI, lr), state, losses
    if state is None:
        state = dict(mU=np.zeros_like(U), vU=np.zeros_like(U), mI=np.zeros_like(I), vI=np.zeros_like(I))
    U1, mU, vU = adam_dense_decay_step(U, state["mU"], state["vU"], gU, t, lr)
    I1, mI, vI = adam_dense_decay_step(I, state["mI"], state["vI"], gI, t, lr)
    return U1, I1, dict(mU=mU, vU=vU, mI=mI, vI=vI), losses


# --------------------------------------------------------------------------------------
# A6: full-catalogue scores + history mask + top-K     MF/model_api.py:62,113
#                                                       MF/train_new_api.py:594-612
# --------------------------------------------------------------------------------------
REC_TYPES = ("main_branch", "main_with_pop", "condition")


def score_matrix(U, I, users, rec_type="main_branch", pop=None, dtype=np.float64):
    """R = U[users] @ I.T  (MF/model_api.py:62); with popularity heads
    (elu(R)+1)*pop[None,:]  (MF/train_new_api.py:601-602 'main_with_pop';
    MF/model_api.py:113 'condition').  'main_branch' returns raw R (:597).
    """
    if rec_type not in REC_TYPES:
        raise NotImplementedError("we have only implement recommendation method: main main+pop condition")
    R = np.asarray(U, dtype=dtype)[users] @ np.asarray(I, dtype=dtype).T
    if rec_type != "main_branch":
        R = elu_plus_one(R) * np.asarray(pop, dtype=dtype)[None, :]
    return R


def apply_history_mask(R, indptr, indices):
    """tf.sparse.add(R, SparseTensor(-inf at (row, train item)))  MF/train_new_api.py:597,603,608.
    ``indptr``/``indices`` is the CSR of the block's users' train items (row r of the block)."""
    R = R.copy()
    for r in range(R.shape[0]):
        R[r, indices[indptr[r]:indptr[r + 1]]] = -np.inf
    return R


def topk_desc_lower_index_first(R, k):
    """tf.nn.top_k(R, k).indices: sorted by score descending, ties -> lower index first [TF-ext].
    np.lexsort sorts by the last key first; secondary key = column index ascending."""
    n = R.shape[1]
    if k > n:
        raise ValueError("input must have at least k columns")   # TF InvalidArgument [TF-ext]
    out = np.empty((R.shape[0], k), dtype=np.int32)
    cols = np.arange(n)
    for r in range(R.shape[0]):
        order = np.lexsort((cols, -R[r]))
        out[r] = order[:k]
    return out


def recommend_topk(U, I, users, indptr, indices, k=50, rec_type="main_branch", pop=None, dtype=np.float64):
    """The three recommendation heads end to end (DatasetApi_Model.do_recommendation,
    MF/train_new_api.py:614-640).  Returns (idx int32[Bu,k], val dtype[Bu,k])."""
    R = apply_history_mask(score_matrix(U, I, users, rec_type, pop, dtype), indptr, indices)
    idx = topk_desc_lower_index_first(R, k)
    return idx, np.take_along_axis(R, idx.astype(np.int64), axis=1)


def merge_partial_topk(vals, idxs, k):
    """Merge R partial lists (per item shard) into one: order (-val, idx).  This is the
    MI355X-side addition; the reference has no counterpart.  vals/idxs: [R,Bu,k]."""
    R, Bu, kk = vals.shape
    v = np.transpose(vals, (1, 0, 2)).reshape(Bu, R * kk)
    i = np.transpose(idxs, (1, 0, 2)).reshape(Bu, R * kk)
    out_i = np.empty((Bu, k), dtype=np.int32)
    out_v = np.empty((Bu, k), dtype=vals.dtype)
    for r in range(Bu):
        order = np.lexsort((i[r], -v[r]))[:k]
        out_i[r], out_v[r] = i[r][order], v[r][order]
    return out_i, out_v


# --------------------------------------------------------------------------------------
# A7: evaluation blocks + mask build                 MF/train_new_api.py:713-739
# --------------------------------------------------------------------------------------
def build_eval_blocks(eval_user_list, train_user_list, block=2048):
    """Split eval users (dict key order) into blocks of ``block`` and build, per block, the
    COO (row-in-block, item) index array exactly as set_evaluate_obj_pre does (:724-739).
    Returns list of (users list, index int64[nnz,2], rows, nnz)."""
    all_users = list(eval_user_list.keys())
    out = []
    for i in range(0, len(all_users), block):
        bu = all_users[i:i + block]
        rows, items = [], []
        for r, u in enumerate(bu):
            m = train_user_list[u]          # KeyError for unseen users under Data2 (plain dict)
            rows.extend([r] * len(m))
            items.extend(m)
        index = np.array([rows, items], dtype=np.int64).T.reshape(-1, 2)
        out.append((bu, index, len(bu), len(rows)))
    return out


# --------------------------------------------------------------------------------------
# A8: ranking metrics                                   MF/used_metric.py:4-80
# --------------------------------------------------------------------------------------
def get_performance(user_pos_test, r, Ks):
    """precision / recall / ndcg / hit_ratio @K from one user's top list ``r``.

    hit vector = isin(r, target) (used_metric.py:65-67); precision = mean(hit[:K]) (:4-18);
    recall = sum(hit[:K])/len(target) (:55-57); ndcg = sum(hit[:K]/log2(2..K+1)) /
    sum_{j<min(len(target),K)} 1/log2(j+2), 0 when the ideal is 0 (:39-52);
    hit = min(1, sum(hit[:K])) (:60-62).  len(target)==0 divides by zero like the reference.
    """
    hit = np.isin(np.asarray(r), np.asarray(list(user_pos_test))).astype(np.float64)
    n_pos = len(user_pos_test)
    prec, rec, ndcg, hr = [], [], [], []
    with np.errstate(divide="ignore", invalid="ignore"):
        for K in Ks:
            assert K >= 1
            h = hit[:K]
            prec.append(np.mean(h))
            rec.append(np.sum(h) / n_pos)
            tp = 1.0 / np.log2(np.arange(2, K + 2))
            dcg_max = tp[:min(n_pos, K)].sum()
            ndcg.append(0.0 if not dcg_max else (h * tp[:h.size]).sum() / dcg_max)
            hr.append(min(1.0, np.sum(h)))
    return {"recall": np.array(rec), "precision": np.array(prec),
            "ndcg": np.array(ndcg), "hit_ratio": np.array(hr)}


def evaluate_topk(topk, users, eval_user_list, Ks):
    """Reduction of MF/train_new_api.py:741-778: per-user metrics summed, divided by tot_user."""
    res = {k: np.zeros(len(Ks)) for k in ("precision", "recall", "ndcg", "hit_ratio")}
    for u, r in zip(users, topk):
        one = get_performance(eval_user_list.get(u, []), r, Ks)
        for k in res:
            res[k] += one[k]
    for k in res:
        res[k] /= len(users)
    return res


# --------------------------------------------------------------------------------------
# Popularity: pop_pre.py:13-57 and MF/train_new_api.py:952-959, 984-990
# --------------------------------------------------------------------------------------
def pop_pre(slot_items, n_item=None):
    """``slot_items[t]`` = list of (item, n_interactions) read from t_{t}.txt.
    pop[t][i] = (cnt+1)/(total_t+n_item) (1/(total+n_item) for absent items), then per-slot
    min-max to [0,1] (pop_pre.py:31-42).  Returns float64 [T, n_item]."""
    if n_item is None:
        n_item = len({it for s in slot_items for it, _ in s})          # pop_pre.py:13-19
    rows = []
    for s in slot_items:
        total = sum(c for _, c in s)
        row = np.full(n_item, 1.0 / (total + n_item))                   # :31
        for it, c in s:
            row[it] = (c + 1.0) / (total + n_item)                      # :35
        rows.append(row)
    pop = np.array(rows)
    for k in range(pop.shape[0]):                                       # :41-42
        pop[k] = (pop[k] - pop[k].min()) / (pop[k].max() - pop[k].min())
    return pop


def popularity_heads(pop_item_all, gamma, coeff=0.5):
    """Test-time popularity vectors (MF/train_new_api.py:954-959) and the train matrix (:988-990).
    pop_item_all: float64 [I, T].  Returns (last_stage[I], linear_pred[I], train_matrix[I,T-1])."""
    last = np.power(pop_item_all[:, -2], gamma)
    lin = pop_item_all[:, -2] + coeff * (pop_item_all[:, -2] - pop_item_all[:, -3])
    lin[np.where(lin <= 0)] = 1e-9
    lin[np.where(lin > 1.0)] = 1.0
    lin = np.power(lin, gamma)
    train = np.power(pop_item_all[:, :-1], gamma)
    return last, lin, train


def xavier_uniform(rows, cols, rng):
    """tf.contrib.layers.xavier_initializer() = U(-l, l), l = sqrt(6/(fan_in+fan_out)) [TF-ext];
    MF/model_api.py:88-92."""
    lim = np.sqrt(6.0 / (rows + cols))
    return rng.uniform(-lim, lim, size=(rows, cols)).astype(np.float32)
