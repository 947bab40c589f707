import ctypes as C, sys, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic, _lib
dev = torch.device('cuda')
W = synthetic.make_workload('c3', dev)
lib = _lib.load()
hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
users = torch.arange(0, 65536, dtype=torch.int32, device=dev)
out = (C.c_ulonglong * 8)()
PR = len(sys.argv) > 1 and sys.argv[1] == "ord"
ops.score_topk_keys(W.U, W.I, users, 50, 1, W.pop_last, hist, impl="v2", prune=PR); torch.cuda.synchronize()
lib.pda_debug_counters(out, 1)
ops.score_topk_keys(W.U, W.I, users, 50, 1, W.pop_last, hist, impl="v2", prune=PR); torch.cuda.synchronize()
lib.pda_debug_counters(out, 1)
w = out[5]
print("waves %d | per wave: hist cycles %.0f  total cycles %.2fM  finalize %.2fM  push(all) %.2fM of which compaction %.2fM | bad-rank %d overflows %d" %
      (w, out[0] / w, out[4] / w / 1e6, out[1] / w / 1e6, out[3] / w / 1e6, out[2] / w / 1e6, out[6], out[7]))
