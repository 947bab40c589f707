"""Round 6: how many 64-item tiles should the exact warm-up score on the shapes that are all fixed cost?  (warm_tiles 1 .. 4 = 64 .. 256 items; fewer
tiles = a shorter warm-up and a later threshold.)  The reference's 2 048-user blocks of C3, 8 192-user blocks, the whole C1 / C2 blocks; the
popularity head, early-terminating (product default) and dense in visiting order.  Keys are compared with warm_tiles = 4."""
import sys, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic
dev = torch.device('cuda')
for wl, Bu in (("c3", 2048), ("c3", 8192), ("c2", 50000), ("c1", 47890)):
    W = synthetic.make_workload(wl, dev)
    hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
    blocks = [torch.arange(s, s + Bu, dtype=torch.int32, device=dev) for s in range(0, min(W.n_users - Bu + 1, 8 * Bu), Bu)]
    for prune, name in ((True, "early-terminating"), ("order", "dense, visiting order")):
        ref = None
        for wt in (4, 3, 2, 1):
            for b in blocks[:2]:
                k = ops.score_topk_keys(W.U, W.I, b, 50, ops.HEAD_POP, W.pop_last, hist, prune=prune, warm_tiles=wt)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            e0.record()
            for _ in range(n):
                for b in blocks:
                    k = ops.score_topk_keys(W.U, W.I, b, 50, ops.HEAD_POP, W.pop_last, hist, prune=prune, warm_tiles=wt)
                    keys = ops.topk_merge(k, want="keys")
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / (n * len(blocks))
            if ref is None:
                ref = keys
            print("%s %6d users %-22s warm_tiles %d: %.3f ms per block (score + merge)  same keys as 4 tiles: %s" % (wl, Bu, name, wt, ms, bool(torch.equal(ref, keys))), flush=True)
