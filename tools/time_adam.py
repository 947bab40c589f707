"""The six-stream dense-decay Adam sweep alone on tables of config 3's size: tools/time_adam.py [n_users=1000000] [n_items=200000] [d=128] [steps=30]"""
import sys, torch
sys.path.insert(0, ".")
from pda_amd import ops
nU, nI, d, steps = [int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 1000000), (2, 200000), (3, 128), (4, 30))]
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(3)
mk = lambda n: [torch.randn(n, d, device=dev, generator=g) * 0.1, torch.zeros(n, d, device=dev), torch.zeros(n, d, device=dev), torch.zeros(n, d, device=dev)]
A, Bt = mk(nU), mk(nI)
tu, ti = ops.adam_touched_bitmaps(nU, nI, dev)
B = 2048
users = torch.randint(0, nU, (B,), device=dev, dtype=torch.int32)
pos = torch.randint(0, nI, (B,), device=dev, dtype=torch.int32)
neg = torch.randint(0, nI, (B,), device=dev, dtype=torch.int32)
def step():
    ops.adam_mark_rows(users, pos, neg, tu, ti)
    ops.adam_dense_sweep3(A[0], A[1], A[2], A[3], tu, Bt[0], Bt[1], Bt[2], Bt[3], ti, 1e-3)
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps): step()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / steps * 1e3
by = 6.0 * (nU + nI) * d * 4
print("six-stream Adam sweep, %d + %d rows x %d: %.1f us per step, %.2f TB/s = %.3f of 8 TB/s" % (nU, nI, d, us, by / us / 1e6, by / us / 1e6 / 8.0))
