"""The reference's 2 048-user block protocol on config 3, kernels only (ids and CSR already in HBM): for rocprofv3 --kernel-trace --stats."""
import sys
import time

import torch

sys.path.insert(0, ".")
from pda_amd import ops, synthetic

dev = torch.device("cuda:0")
W = synthetic.make_workload(sys.argv[1] if len(sys.argv) > 1 else "c3", dev)
hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
blocks = [torch.arange(s, s + 2048, dtype=torch.int32, device=dev) for s in range(0, min(2048 * 40, W.n_users - 2047), 2048)]
for b in blocks[:4]:
    ops.recommend_topk(W.U, W.I, b, 50, ops.HEAD_POP, W.pop_last, hist)
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in blocks:
    ops.recommend_topk(W.U, W.I, b, 50, ops.HEAD_POP, W.pop_last, hist)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / len(blocks)
print("2048-user blocks, queued back to back: %.3f ms per block = %.2f M users/s" % (dt * 1e3, 2048 / dt / 1e6))
t0 = time.perf_counter()
for b in blocks:
    idx, val = ops.recommend_topk(W.U, W.I, b, 50, ops.HEAD_POP, W.pop_last, hist)
    idx.cpu()
dt = (time.perf_counter() - t0) / len(blocks)
print("2048-user blocks, result fetched per block: %.3f ms per block = %.2f M users/s" % (dt * 1e3, 2048 / dt / 1e6))
