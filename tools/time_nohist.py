import sys, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic
dev = torch.device('cuda')
W = synthetic.make_workload('c3', dev)
hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
users = torch.arange(0, 65536, dtype=torch.int32, device=dev)
def t(h, prune):
    k = ops.score_topk_keys(W.U, W.I, users, 50, 1, W.pop_last, h, prune=prune); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): k = ops.score_topk_keys(W.U, W.I, users, 50, 1, W.pop_last, h, prune=prune)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 4
for prune in ("order", False):
    print("prune=%s  with history %.2f ms   without history %.2f ms" % (prune, t(hist, prune), t(None, prune)))
