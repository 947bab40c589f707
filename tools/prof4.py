"""Cycle counters of the profiling build (tools/build_variant.sh prof -DPDA_V4_PROF): one launch of the v4 sweep."""
import os, sys, ctypes as C, torch
sys.path.insert(0, '.')
here = os.path.dirname(os.path.abspath(__file__))
os.environ["PDA_HIP_LIB"] = os.path.join(here, "..", "pda_amd", "csrc", "variants", os.environ.get("PROFLIB", "libpda_hip_prof.so"))
os.environ["PDA_SCORE_KERNEL"] = "v4"
from pda_amd import ops, synthetic, _lib
mode = {"order": "order", "0": False, "1": True}[sys.argv[1] if len(sys.argv) > 1 else "order"]
head = int(sys.argv[2]) if len(sys.argv) > 2 else 1
Bu = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
dev = torch.device('cuda')
W = synthetic.make_workload("c3", dev, n_users=Bu)
hist = None if os.environ.get('NOHIST') else ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
users = torch.arange(Bu, dtype=torch.int32, device=dev)
pop = W.pop_last if head else None
lib = _lib.load()
lib.pda_debug_prof4.argtypes = [C.c_void_p, C.c_int]
ops.score_topk_keys(W.U, W.I, users, 50, head, pop, hist, prune=mode); torch.cuda.synchronize()
out = (C.c_ulonglong * 24)()
lib.pda_debug_prof4(out, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); st = {}
ops.score_topk_keys(W.U, W.I, users, 50, head, pop, hist, prune=mode, stats=st); e1.record(); torch.cuda.synchronize()
lib.pda_debug_prof4(out, 1)
v = list(out)
nm = max(v[13], 1); nr = nm // 2; nl = nm // 2
print("launch %.3f ms; MFMA waves %d; cand/user %.1f" % (e0.elapsed_time(e1), nm, v[9] / Bu))
print("per MFMA wave [kcycles]: total %.0f  wait-landed %.0f  ring-full %.0f  slow path %.0f (%.0f calls, %.0f clamp)  refresh %.0f (%.0f)  pushed %.0f"
      % (v[0] / nm / 1e3, v[1] / nm / 1e3, v[2] / nm / 1e3, v[3] / nm / 1e3, v[4] / nm, v[14] / nm, v[12] / nm / 1e3, v[5] / nm, v[15] / nm))
print("per rescoring wave [kcycles]: total %.0f  idle %.0f  passes %.0f  cand %.0f" % (v[6] / nr / 1e3, v[7] / nr / 1e3, v[8] / nr, v[9] / nr))
print("  rescoring pass [cycles]: ring + loads %.0f  dot %.0f  next pass requested %.0f  history %.0f  append %.0f; inserted per pass %.1f" % tuple([v[k] / max(v[8], 1) for k in (16, 17, 21, 18, 19)] + [v[20] / max(v[8], 1)]))
print("  compactions per pass %.2f, %.0f cycles each" % (v[23] / max(v[8], 1), v[22] / max(v[23], 1)))
nl = max(nm // 4, 1)
print("per loader wave [kcycles] (two per workgroup): wait for a free slot %.0f  issue %.0f  wait for the loads %.0f" % (v[10] / nl / 1e3, v[11] / nl / 1e3, v[12] / nl / 1e3))
