import sys, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic
dev = torch.device('cuda')
W = synthetic.make_workload('c3', dev)
hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
users = torch.arange(0, 8192, dtype=torch.int32, device=dev)
a = ops.topk_merge(ops.score_topk_keys(W.U, W.I, users, 50, 1, W.pop_last, hist, impl="v1"), want="keys")
for prune in ("order", True):
    st = {}
    b = ops.topk_merge(ops.score_topk_keys(W.U, W.I, users, 50, 1, W.pop_last, hist, prune=prune, stats=st), want="keys")
    torch.cuda.synchronize()
    eq = (a == b)
    print(prune, "equal", bool(eq.all()), "bad rows", int((~eq.all(1)).sum()), {k: int(v) if hasattr(v, 'item') else v for k, v in st.items()})
print("pop range", float(W.pop_last.min()), float(W.pop_last.max()), "U std", float(W.U.std()), "I std", float(W.I.std()))
