"""One rank's step time at the shapes the 8-GPU layouts give it: config 3, users x items per rank, dense
sweep in the shard's visiting order and the early-terminating sweep.  usage: shard_scaling.py  (prints the table of profiles/round4_shard_scaling.txt)"""
import os, sys, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic
dev = torch.device("cuda")
W = synthetic.make_workload("c3", dev)
hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
def t(users, I, pop, prune, n=5, warm_tiles=0):
    for _ in range(2):
        ops.score_topk_keys(W.U, I, users, 50, ops.HEAD_POP, pop, hist, prune=prune, warm_tiles=warm_tiles)
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.score_topk_keys(W.U, I, users, 50, ops.HEAD_POP, pop, hist, prune=prune, warm_tiles=warm_tiles)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
print("# config 3 (d = 128, fp32 tables, K = 50, history masked), one MI355X, median of 5 launches; per rank of an 8-GPU job")
print("# warm tiles: 64-item tiles per split scored by the exact warm-up kernel -- 4 on one GPU, 4 / (item shards) per rank (pda_amd.dist: R shards")
print("# warm up R x 64 x tiles items between them).  Early-terminating column: the shard prunes against its OWN K-th value only (no seed exchange).")
print("# layout (user groups x item shards) | users x items per rank | warm tiles | dense ordered ms | speed-up vs 1 GPU | early-terminating ms | speed-up")
base = {}
for name, nu, ni, shards in (("1 GPU", 262144, 200000, 1), ("1 x 8", 262144, 25000, 8), ("2 x 4", 131072, 50000, 4), ("4 x 2", 65536, 100000, 2),
                             ("8 x 1", 32768, 200000, 1)):
    users = torch.arange(nu, dtype=torch.int32, device=dev)
    I = W.I[:ni].contiguous(); pop = W.pop_last[:ni].contiguous()
    for wt in sorted({4, max(1, 4 // shards)}, reverse=True):
        d, e = t(users, I, pop, "order", warm_tiles=wt), t(users, I, pop, True, warm_tiles=wt)
        if not base: base = {"d": d, "e": e}
        print("%-6s | %7d x %6d | %d | %6.2f | %4.1f x | %5.2f | %4.1f x" % (name, nu, ni, wt, d, base["d"] / d, e, base["e"] / e), flush=True)
