"""Host-side tuning of the CPU baseline port (oracle/cpu_baseline.py): slab x chunk sizes of eval_block_blocked against
eval_block_slabbed / eval_block on BASELINE config 3 shapes.  Runs on the GPU box's host cores (no GPU needed)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import cpu_baseline as cb

torch.manual_seed(0)
nI, d = 200_000, 128
U = torch.randn(20000, d) * 0.1
I = torch.randn(nI, d) * 0.1
pop = torch.rand(nI)
users = torch.arange(100, 100 + 2048)
lens = torch.randint(20, 80, (2048,))
rows = torch.repeat_interleave(torch.arange(2048), lens)
cols = torch.randint(0, nI, (int(lens.sum()),))
cores = torch.get_num_threads()
print("threads", cores)
def t(fn, **kw):
    fn(U, I, pop, users, rows, cols, **kw)
    t0 = time.perf_counter(); fn(U, I, pop, users, rows, cols, **kw); fn(U, I, pop, users, rows, cols, **kw)
    return 2 * 2048 / (time.perf_counter() - t0)
print("intraop", t(cb.eval_block))
torch.set_num_threads(1)
print("slabbed 64", t(cb.eval_block_slabbed, threads=cores))
print("slabbed 16", t(cb.eval_block_slabbed, threads=cores, slab=16))
for slab in (16, 32, 64):
    for chunk in (4096, 16384, 65536):
        print("blocked", slab, chunk, t(cb.eval_block_blocked, threads=cores, slab=slab, chunk=chunk))
