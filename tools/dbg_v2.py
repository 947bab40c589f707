import sys, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from pda_amd import ops
from test_gpu_score_topk import make_case, csr
dev = torch.device('cuda')
d, head = 64, 1
rng = np.random.default_rng(100 + d + head)
nU, nI, K = 300, 1999, 50
U, I, pop, hist = make_case(rng, nU, nI, d)
users = rng.permutation(nU)[:173].astype(np.int32)
rows = [hist[u] for u in users]
for variant in ("full", "nohist", "popnz", "nohist_popnz"):
  for ns in (2, 3):
      p = pop.copy()
      if "popnz" in variant: p[p == 0] = 0.5
      h = None
      if "nohist" not in variant:
          ip, ix = csr(rows)
          h = ops.HistoryCSR(torch.from_numpy(ip).to(dev), torch.from_numpy(ix).to(dev), by_user=False)
      a = (torch.from_numpy(U).to(dev), torch.from_numpy(I).to(dev), torch.from_numpy(users).to(dev), K, 1, torch.from_numpy(p).to(dev), h, 0)
      k1 = ops.score_topk_keys(*a, n_splits=ns, impl="v1")
      k2 = ops.score_topk_keys(*a, n_splits=ns, impl="v2")
      print("   equal all splits:", bool(torch.equal(k1,k2)), [bool(torch.equal(k1[q],k2[q])) for q in range(ns)])
      i1, v1 = ops.unpack_keys(k1[ns-1]); i2, v2 = ops.unpack_keys(k2[ns-1])
      bad = [(r, sorted(set(i1[r]) - set(i2[r]))) for r in range(len(users)) if set(i1[r]) != set(i2[r])]
      print(variant, ns, "rows differing:", len(bad), "of", len(users))
      for r, miss in bad[:5]:
          print("  row", r, "missing", miss[:6], "tiles", [m // 32 for m in miss[:6]], "lanes", [m % 32 for m in miss[:6]], "pop", p[miss[:3]], "hist", sorted(rows[r])[:5])
      rr = np.array([r for r, _ in bad]); 
      if len(rr): print("  bad rows mod 32:", np.bincount(rr % 32, minlength=32), " //32:", np.bincount(rr // 32))
