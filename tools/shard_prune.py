"""Early-terminating sweep over R emulated item shards of a bench workload on ONE GPU: item tiles scored with each shard
pruning against its own K-th value vs against the seed (maximum over the shards of the warm-up K-th values); merged lists
compared with the single-shard result.   usage: shard_prune.py [workload=c3] [users=65536] [R=8]"""
import sys, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic
from pda_amd.dist import shard_range
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
Bu = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
R = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda")
W = synthetic.make_workload(wl, dev, n_users=max(Bu, 131072))
hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
users = torch.arange(Bu, dtype=torch.int32, device=dev)
import os
os.environ["PDA_SCORE_KERNEL"] = "v4"
st = {}
ref = ops.topk_merge(ops.score_topk_keys(W.U, W.I, users, 50, ops.HEAD_POP, W.pop_last, hist, prune=True, stats=st), want="keys")
one = float(st["tiles_scored"][0]) / st["tiles_dense"]
print("1 shard: item tiles scored %.4f of the catalogue" % one)
shards = [(lo, hi, W.I[lo:hi].contiguous(), W.pop_last[lo:hi].contiguous()) for lo, hi in (shard_range(W.n_items, r, R) for r in range(R))]
dense_all = ((W.n_items + 31) // 32) * ((Bu + 127) // 128)
for seeded in (False, True):
    if seeded:
        # the ranks' all-reduce MAX, emulated: warm-up of every shard first
        tk, tm = [], []
        for lo, hi, I_s, pop_s in shards:
            ops.score_topk_keys(W.U, I_s, users, 50, ops.HEAD_POP, pop_s, hist, item_offset=lo, prune=True, n_splits=1,
                                seed_reduce=lambda a, b: (tk.append(a.clone()), tm.append(b.clone())), seed_shards=R)
        seed_k, seed_m = torch.stack(tk).max(0).values, torch.stack(tm).min(0).values
    tot, parts = 0.0, []
    for lo, hi, I_s, pop_s in shards:
        st = {}
        kw = {"seed_reduce": (lambda a, b: (a.copy_(seed_k), b.copy_(seed_m))), "seed_shards": R} if seeded else {}
        k = ops.score_topk_keys(W.U, I_s, users, 50, ops.HEAD_POP, pop_s, hist, item_offset=lo, prune=True, n_splits=1, stats=st, **kw)
        tot += float(st["tiles_scored"][0])
        parts.append(ops.topk_merge(k, want="keys"))
    merged = ops.topk_merge(torch.stack(parts), want="keys")
    print("%d shards, %s: item tiles scored %.4f of the catalogue (%.2fx the single shard)  merged == single-shard lists: %s"
          % (R, "seeded" if seeded else "own K-th value only", tot / dense_all, tot / dense_all / one, torch.equal(merged, ref)))
