"""Host-side cost of one evaluation block (python + ctypes + launches), measured with a catalogue so small that the GPU
is never the bottleneck: what a rank pays per block besides its kernels when the item shard gets small (8 GPUs)."""
import sys, time, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic
dev = torch.device('cuda')
W = synthetic.make_workload('c3', dev)
hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
I = W.I[:2048].contiguous(); pop = W.pop_last[:2048].contiguous()
for Bu in (65536, 262144):
    users = torch.arange(0, Bu, dtype=torch.int32, device=dev)
    for _ in range(3):
        k = ops.score_topk_keys(W.U, I, users, 50, 1, pop, hist, prune="order"); r = ops.topk_merge(k, users, hist)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        k = ops.score_topk_keys(W.U, I, users, 50, 1, pop, hist, prune="order"); r = ops.topk_merge(k, users, hist)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("Bu %d: host enqueue %.0f us/block, incl. GPU drain %.0f us/block" % (Bu, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
