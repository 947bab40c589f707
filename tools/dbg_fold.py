import sys, os, torch, numpy as np
sys.path.insert(0, '.')
from pda_amd import ops
dev = torch.device('cuda')
rng = np.random.default_rng(5)
nU, nI, d, K = 200, 3000, int(sys.argv[1]) if len(sys.argv) > 1 else 128, 50
U = torch.from_numpy((rng.standard_normal((nU, d)) * 0.1).astype(np.float32)).to(dev)
I = torch.from_numpy((rng.standard_normal((nI, d)) * 0.1).astype(np.float32)).to(dev)
pop = (rng.uniform(0, 1, nI) ** 0.22).astype(np.float32)
if len(sys.argv) > 2: pop[rng.integers(0, nI, 60)] = 0.0
pop = torch.from_numpy(pop).to(dev)
users = torch.arange(nU, dtype=torch.int32, device=dev)
a = ops.topk_merge(ops.score_topk_keys(U, I, users, K, 1, pop, None, impl="v1"), want="keys")
for prune in ("order", True):
    st = {}
    b = ops.topk_merge(ops.score_topk_keys(U, I, users, K, 1, pop, None, prune=prune, stats=st), want="keys")
    torch.cuda.synchronize()
    eq = (a == b)
    print("prune", prune, "equal", bool(eq.all()), "rows bad", int((~eq.all(1)).sum()), st)
    if not eq.all():
        r = int((~eq.all(1)).nonzero()[0])
        ka, kb = a[r].cpu().numpy(), b[r].cpu().numpy()
        ia, ib = 0xFFFFFFFF - (ka & 0xFFFFFFFF), 0xFFFFFFFF - (kb & 0xFFFFFFFF)
        print(" row", r, "\n  v1", ia[:12], "\n  v3", ib[:12], "\n missing", sorted(set(ia) - set(ib))[:10], "extra", sorted(set(ib) - set(ia))[:10])
        print("  pop of missing", pop[torch.tensor(sorted(set(ia) - set(ib))[:10], device=dev, dtype=torch.long)].cpu().numpy())
