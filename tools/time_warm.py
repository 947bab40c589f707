"""Time the exact warm-up (pda_score_topk4_phase_*, phase 1) alone.  usage: time_warm.py [workload=c3] [users=262144] [dtype=f32]"""
import sys, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic, _lib
from pda_amd._lib import ptr, stream_ptr, check
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
Bu = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
td = torch.bfloat16 if (len(sys.argv) > 3 and sys.argv[3] == "bf16") else torch.float32
dev = torch.device("cuda")
W = synthetic.make_workload(wl, dev, n_users=max(Bu, 131072), table_dtype=td)
hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
users = torch.arange(Bu, dtype=torch.int32, device=dev)
lib = _lib.load()
order = ops.visiting_order(W.I, W.pop_last)
prep = ops.item_prep4(W.I, W.pop_last, order)
ns = lib.pda_score_topk4_auto_splits(Bu, W.n_items, W.d)
out = torch.empty((ns, Bu, 50), dtype=torch.int64, device=dev)
ws = torch.empty(lib.pda_score_topk4_workspace_bytes(Bu, W.n_items, W.d, ns), dtype=torch.uint8, device=dev)
fn = lib.pda_score_topk4_phase_bf16 if td == torch.bfloat16 else lib.pda_score_topk4_phase_f32
def run():
    check(fn(ptr(W.U), ptr(W.I), ptr(prep), ptr(W.pop_last), ptr(users), Bu, 0, W.n_items, W.d, ptr(hist.indptr), ptr(hist.indices), hist.mode,
             50, 1, 1, ns, 1, 0, None, ptr(out), ptr(ws), stream_ptr()), "phase 1")
for _ in range(3): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): run()
e.record(); torch.cuda.synchronize()
print("%s warm-up of %d users, %d splits: %.3f ms" % (wl, Bu, ns, s.elapsed_time(e) / 10))
