"""bf16-table eval timing: C3 tables rounded to bf16, and a config-5-shaped shard (250k items x d=256, one of 8 ranks)."""
import sys, time, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic
dev = torch.device('cuda')
def run(U, I, pop, hist, users, prune, n=3):
    st = {}
    k = ops.score_topk_keys(U, I, users, 50, 1, pop, hist, prune=prune, stats=st); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        k = ops.score_topk_keys(U, I, users, 50, 1, pop, hist, prune=prune, stats=st)
    e1.record(); torch.cuda.synchronize()
    frac = float(st["tiles_scored"][0]) / st["tiles_dense"] if "tiles_scored" in st else 1.0
    return e0.elapsed_time(e1) / n, frac
W = synthetic.make_workload('c3', dev)
hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
users = torch.arange(0, 65536, dtype=torch.int32, device=dev)
Ub, Ib = W.U.bfloat16(), W.I.bfloat16()
for name, (U, I) in (("f32 ", (W.U, W.I)), ("bf16", (Ub, Ib))):
    d_ms, _ = run(U, I, W.pop_last, hist, users, False)
    do_ms, _ = run(U, I, W.pop_last, hist, users, "order")
    o_ms, fr = run(U, I, W.pop_last, hist, users, True)
    print("C3 %s tables: dense natural %.2f ms, dense ordered %.2f ms (%.2fM users/s)  early-stop %.3f ms (%.1fM users/s, %.3f of tiles)" % (name, d_ms, do_ms, 65536 / do_ms / 1e3, o_ms, 65536 / o_ms / 1e3, fr))
# config-5 shard: d=256, 250k local items, users drawn from 1M rows (the user table of 10M x 256 bf16 = 5 GB is replicated; 1M here)
g = torch.Generator(device=dev); g.manual_seed(5)
U5 = (torch.randn(1_000_000, 256, device=dev, generator=g) * 0.07).bfloat16()
I5 = (torch.randn(250_000, 256, device=dev, generator=g) * 0.07).bfloat16()
pop5 = W.pop_last.repeat(2)[:250_000].contiguous()
d_ms, _ = run(U5, I5, pop5, None, users, "order")
o_ms, fr = run(U5, I5, pop5, None, users, True)
fl = 2.0 * 65536 * 250000 * 256
print("C5 shard (250k x 256 bf16): dense %.2f ms = %.0f TFLOP/s bf16 (%.1f%% of 2.5 PF), %.2fM users/s/rank; ordered %.3f ms (%.3f of tiles)"
      % (d_ms, fl / d_ms / 1e9, fl / d_ms / 1e9 / 25.0, 65536 / d_ms / 1e3, o_ms, fr))
