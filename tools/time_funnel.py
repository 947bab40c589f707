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
# measurements: FUNNEL_TUNE="fail_p,growth,cap_e,first_tiles"  FUNNEL_TUNE2="first_mult,late_den,late_growth_x10"
import ctypes as C
from pda_amd import _lib
L = _lib.load()
if os.environ.get("FUNNEL_TUNE"):
    a = os.environ["FUNNEL_TUNE"].split(",")
    L.pda_debug_funnel_tune.argtypes = [C.c_double, C.c_int, C.c_int, C.c_int]
    L.pda_debug_funnel_tune(float(a[0]), int(a[1]), int(a[2]), int(a[3]))
if os.environ.get("FUNNEL_TUNE2"):
    a = [int(x) for x in os.environ["FUNNEL_TUNE2"].split(",")]
    L.pda_debug_funnel_tune2(a[0], a[1], a[2])
out = (C.c_int * 96)()
ns = L.pda_debug_funnel_schedule_for(Bu, W.n_items, W.d, 50, out, 32)
print("schedule", [(out[3 * i], out[3 * i + 1], out[3 * i + 2]) for i in range(ns)])
st = {}
for _ in range(3):          # (the workspace of a call is allocated per call: the allocator's blocks exist after two)
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
