"""Early-terminating sweep over R emulated item shards of a bench workload on ONE GPU (one thread per shard, all-reduces among
the threads -- what pda_amd.dist does over RCCL): item tiles scored with each shard pruning against its own K-th value, against
the plain seed, and against the seed tightened by PDA_SEED_ROUNDS bisection rounds; merged lists compared with one shard's.
usage: shard_prune.py [workload=c3] [users=65536] [R=8]"""
import os, sys, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from pda_amd import ops, synthetic
from pda_amd.dist import shard_range
from test_gpu_score_topk import run_emulated_shards
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
Bu = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
R = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda")
W = synthetic.make_workload(wl, dev, n_users=max(Bu, 131072))
hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
users = torch.arange(Bu, dtype=torch.int32, device=dev)
os.environ["PDA_SCORE_KERNEL"] = "v4"
st = {}
ref = ops.topk_merge(ops.score_topk_keys(W.U, W.I, users, 50, ops.HEAD_POP, W.pop_last, hist, prune=True, stats=st), want="keys")
one = float(st["tiles_scored"][0]) / st["tiles_dense"]
print("1 shard: item tiles scored %.4f of the catalogue" % one)
shards = [(lo, W.I[lo:hi].contiguous(), W.pop_last[lo:hi].contiguous()) for lo, hi in (shard_range(W.n_items, r, R) for r in range(R))]
dense_all = ((W.n_items + 31) // 32) * ((Bu + 127) // 128)
for name, seeded, rounds in (("own K-th value only", False, 0), ("plain seed", True, 0), ("seed + 2 rounds", True, 2), ("seed + 3 rounds", True, 3),
                             ("seed + 5 rounds", True, 5)):
    os.environ["PDA_SEED_ROUNDS"] = str(rounds)
    def fn(r, coll):
        lo, I_s, pop_s = shards[r]
        s2 = {}
        kw = {}
        if seeded:
            kw = {"seed_reduce": lambda b: coll.all_reduce(r, b, "max"),          # float32 [3, Bu]: (K-th, ceil(K / R)-th, minus the latter)
                  "seed_sum": lambda c: coll.all_reduce(r, c, "sum"), "seed_shards": R}
        k = ops.score_topk_keys(W.U, I_s, users, 50, ops.HEAD_POP, pop_s, hist, item_offset=lo, prune=True, n_splits=1, stats=s2, **kw)
        return ops.topk_merge(k, want="keys"), float(s2["tiles_scored"][0])
    res = run_emulated_shards(R, fn) if seeded else [fn(r, None) for r in range(R)]      # (first pass sequential: warms ops' caches)
    merged = ops.topk_merge(torch.stack([p for p, _ in res]), want="keys")
    tot = sum(t for _, t in res)
    print("%d shards, %-20s item tiles scored %.4f of the catalogue (%.2fx one shard)  merged == one-shard lists: %s"
          % (R, name + ":", tot / dense_all, tot / dense_all / one, torch.equal(merged, ref)))
