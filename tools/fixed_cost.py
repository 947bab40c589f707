"""Fixed per-block cost of the ordered sweep kernel: catalogues of 64 .. 4096 items (C3 tables, 65 536 users)."""
import sys, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic
dev = torch.device('cuda')
W = synthetic.make_workload('c3', dev)
hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
users = torch.arange(0, 65536, dtype=torch.int32, device=dev)
for n in (64, 128, 256, 512, 1024, 2048, 4096):
    I = W.I[:n].contiguous(); pop = W.pop_last[:n].contiguous()
    for h in (hist, None):
        for _ in range(2): k = ops.score_topk_keys(W.U, I, users, 50, 1, pop, h, prune="order", n_splits=1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): k = ops.score_topk_keys(W.U, I, users, 50, 1, pop, h, prune="order", n_splits=1)
        e1.record(); torch.cuda.synchronize()
        print("items %5d %s: %.0f us per 65536-user block" % (n, "hist  " if h is not None else "nohist", e0.elapsed_time(e1) / 10 * 1e3))
