import sys, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic
dev = torch.device('cuda')
W = synthetic.make_workload('c3', dev)
hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
users = torch.arange(0, 65536, dtype=torch.int32, device=dev)
for prune in ("order", True, False):
    for h in (hist, None):
        st = {}
        k = ops.score_topk_keys(W.U, W.I, users, 50, 1, W.pop_last, h, prune=prune, stats=st); torch.cuda.synchronize()
        print(prune, "hist" if h is not None else "nohist", {a: (int(b) if hasattr(b, 'item') else b) for a, b in st.items()}, "pairs/user %.1f" % (int(st['pairs_rescored']) / 65536))
