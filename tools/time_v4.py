"""A/B at a bench workload: generation 3 vs generation 4 of the pre-filtered kernel in the three sweep modes; keys compared.
usage: time_v4.py [workload=c3] [users per block=65536] [heads=10] [kernels=v3,v4]"""
import os, sys, time, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic
wl = sys.argv[1] if len(sys.argv) > 1 else 'c3'
Bu = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
heads = [int(c) for c in (sys.argv[3] if len(sys.argv) > 3 else "10")]
kernels = (sys.argv[4] if len(sys.argv) > 4 else "v3,v4").split(",")
dev = torch.device('cuda')
tdt = torch.bfloat16 if (len(sys.argv) > 5 and sys.argv[5] == 'bf16') else torch.float32
nus = int(sys.argv[6]) if len(sys.argv) > 6 else None
nit = int(sys.argv[7]) if len(sys.argv) > 7 else None
mh = int(sys.argv[8]) if len(sys.argv) > 8 else None
W = synthetic.make_workload(wl, dev, table_dtype=tdt, n_users=nus, n_items=nit, mean_hist=mh)
hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
Bu = min(Bu, W.n_users)
blocks = [torch.arange(s, s + Bu, dtype=torch.int32, device=dev) for s in range(0, min(W.n_users - Bu + 1, 4 * Bu), Bu)]
def run(kern, prune, head, n=3):
    os.environ["PDA_SCORE_KERNEL"] = kern
    pop = W.pop_last if head else None
    st = {}
    k = ops.score_topk_keys(W.U, W.I, blocks[0], 50, head, pop, hist, prune=prune, stats=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        for b in blocks:
            k = ops.score_topk_keys(W.U, W.I, b, 50, head, pop, hist, prune=prune, stats=st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (n * len(blocks))
    frac = float(st["tiles_scored"][0]) / st["tiles_dense"] if "tiles_scored" in st else 1.0
    cand = float(st["pairs_rescored"][0]) / Bu if "pairs_rescored" in st else 0.0
    return ms, frac, cand, ops.topk_merge(k, want="keys")
for head in heads:
    for prune, name in ((("order", "dense ordered"),) if os.environ.get("ONLY_ORDER") else (("order", "dense ordered"), (False, "dense natural"), (True, "early stop"))):
        ref = None
        for kern in kernels:
            ms, frac, cand, keys = run(kern, prune, head)
            same = "" if ref is None else " same=%s" % torch.equal(ref, keys)
            ref = keys if ref is None else ref
            fl = 2.0 * Bu * W.n_items * W.d / (ms * 1e-3) / 1e12
            print("head=%d %-14s %s: %.3f ms  %.2f M users/s  %.0f TF (%.3f of 2.5 PF)  tiles %.4f  cand/user %.0f%s"
                  % (head, name, kern, ms, Bu / ms / 1e3, fl, fl / 2500, frac, cand, same), flush=True)
