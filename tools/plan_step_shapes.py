import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from pda_amd import ops
dev = torch.device("cuda")
B, d, nU, nI = 32768, 64, 50000, 20000
U = torch.randn(nU, d, device=dev) * 0.1
I = torch.randn(nI, d, device=dev) * 0.1
g = torch.Generator(device="cpu"); g.manual_seed(1)
def mk(kind):
    users = torch.randperm(nU, generator=g)[:B].to(torch.int32)
    if kind == "uniform":
        pos = torch.randint(0, nI, (B,), generator=g, dtype=torch.int32); neg = torch.randint(0, nI, (B,), generator=g, dtype=torch.int32)
    elif kind == "distinct":
        pos = (torch.arange(B) % nI).to(torch.int32); neg = ((torch.arange(B) + 7777) % nI).to(torch.int32)
    elif kind == "zipf":
        w = 1.0 / torch.arange(1, nI + 1, dtype=torch.float64); pos = torch.multinomial(w, B, replacement=True, generator=g).to(torch.int32)
        neg = torch.randint(0, nI, (B,), generator=g, dtype=torch.int32)
    elif kind == "zipf_shuffled_ids":
        w = 1.0 / torch.arange(1, nI + 1, dtype=torch.float64); perm = torch.randperm(nI, generator=g)
        pos = perm[torch.multinomial(w, B, replacement=True, generator=g)].to(torch.int32)
        neg = torch.randint(0, nI, (B,), generator=g, dtype=torch.int32)
    return users.to(dev), pos.to(dev), neg.to(dev)
for kind in ("uniform", "distinct", "zipf", "zipf_shuffled_ids"):
    u, p, n = mk(kind)
    plan = ops.triplet_plan(u, p, n)[0]
    hdr = ops.plan_header(plan)
    sc = [None]
    def body():
        sc[0] = ops.bpr_step_plan(U, I, u, p, n, regs=1e-2, reg_div=B, lr=1e-3, plan=plan, scratch=sc[0])
    for _ in range(3): body()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): body()
    e1.record(); torch.cuda.synchronize()
    cnt = torch.bincount(torch.cat([p, n]).long(), minlength=nI)
    print("%-18s segments %6d  longest %5d  segments > 8: %5d   %.1f us per step" % (kind, hdr[0], int(cnt.max()), int((cnt > 8).sum()), e0.elapsed_time(e1) / 20 * 1e3))
