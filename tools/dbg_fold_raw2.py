import sys, torch, numpy as np
sys.path.insert(0, '.')
from pda_amd import ops
dev = torch.device('cuda')
rng = np.random.default_rng(5)
d, nI, nU, K = 128, 1999, 300, 50
U = torch.from_numpy((rng.standard_normal((nU, d)) * 0.1).astype(np.float32)).to(dev)
I = torch.from_numpy((rng.standard_normal((nI, d)) * 0.1).astype(np.float32)).to(dev)
users = torch.arange(nU, dtype=torch.int32, device=dev)
a = ops.topk_merge(ops.score_topk_keys(U, I, users, K, 0, None, None, impl="v1"), want="keys")
b = ops.topk_merge(ops.score_topk_keys(U, I, users, K, 0, None, None, prune=False), want="keys")
torch.cuda.synchronize()
eq = (a == b)
b2 = ops.topk_merge(ops.score_topk_keys(U, I, users, K, 0, None, None, prune=False), want="keys")
print("bad rows", int((~eq.all(1)).sum()), "second run same:", bool((b == b2).all()))
S = (U.double() @ I.double().T).cpu().numpy()
rows = (~eq.all(1)).nonzero().flatten().tolist()
print('bad rows', rows)
for r in rows[:40]:
    ka, kb = a[r].cpu().numpy(), b[r].cpu().numpy()
    ia, ib = (0xFFFFFFFF - (ka & 0xFFFFFFFF)).astype(np.int64), (0xFFFFFFFF - (kb & 0xFFFFFFFF)).astype(np.int64)
    miss, extra = sorted(set(ia) - set(ib)), sorted(set(ib) - set(ia))
    srt = np.sort(S[r])[::-1]
    print("row", r, "tiles", [m // 32 for m in miss], "missing", miss, "rank", [int((S[r] > S[r, m]).sum()) for m in miss], "score", [round(float(S[r, m]), 4) for m in miss], "kth", round(float(srt[K - 1]), 4), "extra", extra)
