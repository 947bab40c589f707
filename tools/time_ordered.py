"""A/B: dense v2 sweep vs ordered sweep with early termination at a bench workload (default C3)."""
import sys, time, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic
wl = sys.argv[1] if len(sys.argv) > 1 else 'c3'
Bu = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device('cuda')
W = synthetic.make_workload(wl, dev)
hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
Bu = min(Bu, W.n_users)
blocks = [torch.arange(s, s + Bu, dtype=torch.int32, device=dev) for s in range(0, min(W.n_users - Bu + 1, 8 * Bu), Bu)]
def run(prune, head, n=3):
    pop = W.pop_last if head else None
    st = {}
    t0 = time.perf_counter()
    k = ops.score_topk_keys(W.U, W.I, blocks[0], 50, head, pop, hist, prune=prune, stats=st, n_splits=NS if prune else 0)   # warm-up incl. prep / hist reorder
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        for b in blocks:
            k = ops.score_topk_keys(W.U, W.I, b, 50, head, pop, hist, prune=prune, stats=st, n_splits=NS if prune else 0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (n * len(blocks))
    frac = float(st["tiles_scored"][0]) / st["tiles_dense"] if "tiles_scored" in st else 1.0
    global CAND
    CAND = float(st["pairs_rescored"][0]) / Bu if "pairs_rescored" in st else 0.0
    return ms, frac, first, k
for head in (1, 0):
    a = run(False, head); b = run(True, head)
    same = torch.equal(ops.topk_merge(a[3], want="keys"), ops.topk_merge(b[3], want="keys"))
    print("cand/user %.0f " % CAND, end="")
    print("head=%d  dense %.3f ms  ordered %.3f ms (tiles scored %.4f of dense; first call incl. prep+hist reorder %.1f ms)  same=%s  -> %.2fM users/s"
          % (head, a[0], b[0], b[1], b[2] * 1e3, same, Bu / b[0] / 1e3))
