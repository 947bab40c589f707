"""Effective shader clock during the score kernel: per-wave cycle counter (s_memtime) total / kernel wall time.  Needs -DPDA_ABLATION."""
import ctypes as C, os, sys, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic, _lib
dev = torch.device('cuda')
W = synthetic.make_workload('c3', dev)
lib = _lib.load()
hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
users = torch.arange(0, 65536, dtype=torch.int32, device=dev)
out = (C.c_ulonglong * 8)()
for abl in sys.argv[1:]:
    os.environ["PDA_ABLATE"] = abl
    for _ in range(2):
        ops.score_topk_keys(W.U, W.I, users, 50, 1, W.pop_last, hist, impl="v2", prune=False)
    torch.cuda.synchronize(); lib.pda_debug_counters(out, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        ops.score_topk_keys(W.U, W.I, users, 50, 1, W.pop_last, hist, impl="v2", prune=False)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 4
    lib.pda_debug_counters(out, 1)
    cyc = out[4] / out[5]
    print("ABL=%s  %.2f ms/launch  %.2fM cycles per wave  -> %.2f GHz if a wave lives the whole launch" % (abl, ms, cyc / 1e6, cyc / ms / 1e6))
