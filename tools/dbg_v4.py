import os, sys, torch, numpy as np
sys.path.insert(0, '.')
from pda_amd import ops
dev = torch.device('cuda')
def case(d, head, mode, ns, nU=300, nI=1999, hist=True, K=50, seed=1):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    U = torch.randn(nU, d, device=dev, generator=g) * 0.1
    I = torch.randn(nI, d, device=dev, generator=g) * 0.1
    pop = torch.rand(nI, device=dev, generator=g) ** 0.22
    users = torch.arange(nU, dtype=torch.int32, device=dev)
    h = None
    if hist:
        rows = [np.random.default_rng(u).integers(0, nI, 20) for u in range(nU)]
        h = ops.HistoryCSR.from_lists(rows, dev, by_user=True)
    os.environ["PDA_SCORE_KERNEL"] = "v4"
    a = ops.topk_merge(ops.score_topk_keys(U, I, users, K, head, pop if head else None, h, impl="v1"), want="keys")
    st = {}
    kb = ops.score_topk_keys(U, I, users, K, head, pop if head else None, h, prune=mode, n_splits=ns, stats=st)
    b = ops.topk_merge(kb, want="keys")
    torch.cuda.synchronize()
    ok = torch.equal(a, b)
    msg = ""
    if not ok:
        bad = (a != b).any(1).nonzero().flatten()
        ai, av = ops.unpack_keys(a); bi, bv = ops.unpack_keys(b)
        r = int(bad[0])
        miss = sorted(set(ai[r]) - set(bi[r])); extra = sorted(set(bi[r]) - set(ai[r]))
        msg = " bad rows %d first %d missing %s extra %s" % (len(bad), r, miss[:6], extra[:6])
    print("d=%d head=%d mode=%s ns=%d hist=%d: %s%s cand=%d" % (d, head, mode, ns, hist, "OK" if ok else "MISMATCH", msg, int(st["pairs_rescored"][0])), flush=True)
import itertools
for d, nU, hist in itertools.product((64, 128), (173, 300), (False, True)):
    case(d, 0, False, 0, nU=nU, hist=hist)
    case(d, 0, False, 1, nU=nU, hist=hist)
for head in (0, 1):
    for mode in (False, "order"):
        case(256, head, mode, 1, hist=False, nU=100)
        case(256, head, mode, 1, hist=False, nU=300, nI=600)
