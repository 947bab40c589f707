import sys, torch, numpy as np
sys.path.insert(0, '.')
from pda_amd import ops
dev = torch.device('cuda')
rng = np.random.default_rng(5)
for d in [int(x) for x in sys.argv[1].split(",")]:
    for nI in [int(x) for x in sys.argv[2].split(",")]:
        nU, K = 300, 50
        U = torch.from_numpy((rng.standard_normal((nU, d)) * 0.1).astype(np.float32)).to(dev)
        I = torch.from_numpy((rng.standard_normal((nI, d)) * 0.1).astype(np.float32)).to(dev)
        users = torch.arange(nU, dtype=torch.int32, device=dev)
        a = ops.topk_merge(ops.score_topk_keys(U, I, users, K, 0, None, None, impl="v1"), want="keys")
        st = {}
        b = ops.topk_merge(ops.score_topk_keys(U, I, users, K, 0, None, None, prune=False, stats=st), want="keys")
        torch.cuda.synchronize()
        eq = (a == b)
        print("d", d, "nI", nI, "equal", bool(eq.all()), "rows bad", int((~eq.all(1)).sum()), {k: int(v) if hasattr(v, 'item') else v for k, v in st.items()})
