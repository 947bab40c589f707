"""The funnel alone on a bench workload (for rocprofv3): tools/time_funnel.py [workload=c3] [users=262144] [calls=6]"""
import os, sys, torch
sys.path.insert(0, ".")
from pda_amd import ops, synthetic
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
dev = torch.device("cuda")
W = synthetic.make_workload(wl, dev)
Bu = min(int(sys.argv[2]) if len(sys.argv) > 2 else 262144, W.n_users)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
users = torch.arange(Bu, dtype=torch.int32, device=dev)
hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
os.environ["PDA_SCORE_FUNNEL"] = "1"
st = {}
ops.score_topk_keys(W.U, W.I, users, 50, ops.HEAD_RAW, None, hist, stats=st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    ops.score_topk_keys(W.U, W.I, users, 50, ops.HEAD_RAW, None, hist, stats=st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print("%s %d users, raw head through the funnel: %.3f ms per call = %.3f of 2.5 PF; exact rescorings per user %.1f, rows through the fallback %d, kernel %s" %
      (wl, Bu, ms, 2.0 * Bu * W.n_items * W.d / (ms * 1e-3) / 2.5e15, float(st["pairs_rescored"][0]) / Bu, int(st["fallback_rows"][0]), ops.kernel_identity(st["kernel_id"][0])))
