"""One rank's step of the replicated-hot-items path (pda_amd.dist, round 4) at the shapes of config 3 on R item shards, timed on one
GPU: hot pass of this rank's 1 / R of the users on the 256 replicated rows + K-th values, cold sweep of the whole block against the
rank's shard (without its hot rows) from empty lists against the seed, split merge, id remap.  The two collectives (4 bytes per user
all-gathered; the all-to-all of the lists, which the pipeline hides under the next block) are not in it.
usage: hot_items.py [R=8] [users per step=262144] [order|stop]"""
import sys, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic
from pda_amd.dist import ItemShardedTopK, shard_range
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Bu = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
MODE = True if (len(sys.argv) > 3 and sys.argv[3] == "stop") else "order"      # "stop": the early-terminating sweep (the product default)
dev = torch.device("cuda")
W = synthetic.make_workload("c3", dev)
hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
users = torch.arange(Bu, dtype=torch.int32, device=dev)
K, H, nI = 50, 256, W.n_items
pop = W.pop_last
hot_ids = torch.argsort(pop, descending=True, stable=True)[:H].sort().values
rows = torch.repeat_interleave(torch.arange(W.n_users, device=dev), hist.indptr[1:] - hist.indptr[:-1])
idx = hist.indices.long()
pos = torch.searchsorted(hot_ids, idx).clamp_(max=H - 1)
is_hot = hot_ids[pos] == idx
def csr(sel, local):
    ptr = torch.zeros(W.n_users + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.bincount(rows[sel], minlength=W.n_users), 0, out=ptr[1:])
    return ops.HistoryCSR(ptr, local.to(torch.int32).contiguous(), by_user=True)
h_hot = csr(is_hot, pos[is_hot])
hot_I, hot_pop, hot_gid = W.I[hot_ids].contiguous(), pop[hot_ids].contiguous(), hot_ids.to(torch.int32)
def med(f, n=7):
    for _ in range(2): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
# the seed of ALL users (what the all-gather delivers), and the one-GPU step it is compared with
seed = ops.kth_value(ops.score_topk_keys(W.U, hot_I, users, K, 1, hot_pop, h_hot, 0, 1, prune="order"), K - 1)
one = med(lambda: ops.score_topk_keys(W.U, W.I, users, K, 1, pop, hist, prune=MODE))
ref = ops.topk_merge(ops.score_topk_keys(W.U, W.I, users, K, 1, pop, hist, prune="order"), want="keys")
print("# config 3, %d users per step, %d item shards, %d replicated hot rows, %s sweep; one MI355X, median of 7; ms" % (Bu, R, H, "early-terminating" if MODE is True else "dense"))
print("# one GPU (whole catalogue, one call): %.3f ms" % one)
lists = [ItemShardedTopK.remap_keys(ops.score_topk_keys(W.U, hot_I, users, K, 1, hot_pop, h_hot, 0, 1, prune="order")[0], hot_gid)]
worst = 0.0
for r in range(R):
    lo, hi = shard_range(nI, r, R)
    mine = (hot_ids >= lo) & (hot_ids < hi)
    cm = torch.ones(hi - lo, dtype=torch.bool, device=dev); cm[hot_ids[mine] - lo] = False
    cl = torch.nonzero(cm).flatten(); cio = torch.cumsum(cm, 0) - 1
    loc = idx - lo; is_cold = (loc >= 0) & (loc < hi - lo) & ~is_hot
    h_cold = csr(is_cold, cio[loc[is_cold]])
    I_cold, pop_cold, cold_gid = W.I[lo:hi][cl].contiguous(), pop[lo:hi][cl].contiguous(), (cl + lo).to(torch.int32)
    per = Bu // R
    slice_u = users[r * per:(r + 1) * per].contiguous()
    def hot_pass():
        k = ops.score_topk_keys(W.U, hot_I, slice_u, K, 1, hot_pop, h_hot, 0, 1, prune="order")
        return ItemShardedTopK.remap_keys(k[0], hot_gid), ops.kth_value(k, K - 1)
    def cold_pass():
        k = ops.sweep_from_seed(W.U, I_cold, users, K, 1, pop_cold, h_cold, 0, seed, prune=MODE)
        return ItemShardedTopK.remap_keys(k[0] if k.shape[0] == 1 else ops.topk_merge(k, want="keys"), cold_gid)
    def sweep_only():
        return ops.sweep_from_seed(W.U, I_cold, users, K, 1, pop_cold, h_cold, 0, seed, prune=MODE)
    th, tc, ts = med(hot_pass), med(cold_pass), med(sweep_only)
    worst = max(worst, th + tc)
    lists.append(cold_pass())
    print("rank %d: %6d cold items | hot pass %.3f + cold pass %.3f (sweep call %.3f, split merge + id remap %.3f) = %.3f ms -> %.2f x one GPU" %
          (r, I_cold.shape[0], th, tc, ts, tc - ts, th + tc, one / (th + tc)), flush=True)
got = ops.topk_merge(torch.stack(lists).contiguous(), want="keys")
print("# merged lists of the %d shards + the hot list equal the one-GPU lists: %s" % (R, bool(torch.equal(got, ref))))
print("# slowest rank %.3f ms: %d shards = %.2f x one GPU (collectives not included: 4 B per user all-gathered before the sweep, the all-to-all of the lists under the next block)" % (worst, R, one / worst))
