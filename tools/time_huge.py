"""The huge geometry against another one (lds | many) on a bench workload: time per block, loop entries, candidates (HIP events around the call).
usage: time_huge.py [workload=c3] [users per block=262144] [geometries=lds,huge] [n_items override: the fixed cost of a call, 0 = none] [item splits, 0 = the library's choice]"""
import os, sys, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic
wl = sys.argv[1] if len(sys.argv) > 1 else 'c3'
Bu = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
geos = (sys.argv[3] if len(sys.argv) > 3 else "lds,huge").split(",")
dev = torch.device('cuda')
W = synthetic.make_workload(wl, dev, n_items=(int(sys.argv[4]) or None) if len(sys.argv) > 4 else None)
NS = int(sys.argv[5]) if len(sys.argv) > 5 else 0
WT = int(sys.argv[6]) if len(sys.argv) > 6 else 0
hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
Bu = min(Bu, W.n_users)
blocks = [torch.arange(s, s + Bu, dtype=torch.int32, device=dev) for s in range(0, min(W.n_users - Bu + 1, 3 * Bu), Bu)]
os.environ["PDA_SCORE_KERNEL"] = "v4"
ref = None
for geo in geos:
    if geo == "auto":                    # the library's own choice of kernel, geometry and item splits
        os.environ.pop("PDA_SCORE_LISTS", None)
        os.environ.pop("PDA_SCORE_KERNEL", None)
    else:
        os.environ["PDA_SCORE_LISTS"] = geo
    st = {}
    k = ops.score_topk_keys(W.U, W.I, blocks[0], 50, ops.HEAD_POP, W.pop_last, hist, prune="order", stats=st, n_splits=NS, warm_tiles=WT)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 4
    e0.record()
    for i in range(n):
        for b in blocks:
            k = ops.score_topk_keys(W.U, W.I, b, 50, ops.HEAD_POP, W.pop_last, hist, prune="order", stats=st, n_splits=NS, warm_tiles=WT)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (n * len(blocks))
    keys = ops.topk_merge(k, want="keys")
    same = "" if ref is None else " same keys=%s" % torch.equal(ref, keys)
    ref = keys if ref is None else ref
    fl = 2.0 * Bu * W.n_items * W.d / (ms * 1e-3) / 1e12
    wgs = -(-Bu // 1024) * k.shape[0]
    print("%s %d users %-6s splits %d: %.3f ms  %.0f TF (%.3f of 2.5 PF)  cand/user %.3f  loop entries per wave %.2f  error %d%s" %
          (wl, Bu, geo, k.shape[0], ms, fl, fl / 2500, float(st["pairs_rescored"][0]) / Bu, float(st["huge_entries"][0]) / (4 * wgs), int(st["error"][0]), same), flush=True)
