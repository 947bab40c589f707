"""Soak test of the funnel: random shapes, K, dtypes, masks and item shards -- the funnel's packed keys against the other kernels' (generation 4 / 3, forced) on the
same inputs, bit for bit.  tools/soak_funnel.py [cases=150] [seed=1]"""
import os, sys
import torch
sys.path.insert(0, ".")
from pda_amd import ops
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(seed)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
bad = fb = 0
for c in range(n_cases):
    d = (64, 128, 256)[ri(0, 2)]
    nI, nu, K = ri(4096, 90000), ri(1, 9000), ri(1, 54)
    bf = ri(0, 3) == 0
    scale = (0.02, 0.1, 1.0, 30.0)[ri(0, 3)]
    U = torch.randn(nu + 50, d, generator=g) * scale
    I = torch.randn(nI, d, generator=g) * scale * (0.2 + 1.6 * torch.rand(nI, 1, generator=g))
    if ri(0, 4) == 0:
        I[torch.randint(0, nI, (nI // 3,), generator=g)] = 0.0          # exact ties at 0
    if ri(0, 4) == 0:
        U[torch.randint(0, nu, (max(1, nu // 10),), generator=g)] = 0.0
    U, I = U.to(dev), I.to(dev)
    if bf:
        U, I = U.bfloat16(), I.bfloat16()
    users = torch.randperm(nu + 50, generator=g)[:nu].to(torch.int32).to(dev)
    hist = None
    if ri(0, 2) > 0:
        # (vectorised: 120 sorted draws per user, the first len kept -- a row may name an item twice, which the binary searches do not mind)
        lens = torch.randint(0, 120, (nu + 50,), generator=g)
        indptr = torch.zeros(nu + 51, dtype=torch.int64); indptr[1:] = torch.cumsum(lens, 0)
        draws = torch.randint(0, nI, (nu + 50, 120), generator=g)
        keep = torch.arange(120)[None, :] < lens[:, None]
        draws = torch.where(keep, draws, torch.full_like(draws, nI + 1)).sort(dim=1).values
        idx = draws[keep.sort(dim=1, descending=True).values].to(torch.int32)
        hist = ops.HistoryCSR(indptr.to(dev), idx.to(dev), by_user=True)
    lo = ri(0, nI // 3) if ri(0, 2) == 0 else 0
    Ish = I[lo:].contiguous()
    if Ish.shape[0] < 4096:
        continue
    os.environ["PDA_SCORE_FUNNEL"] = "1"
    st = {}
    kf = ops.topk_merge(ops.score_topk_keys(U, Ish, users, K, ops.HEAD_RAW, None, hist, item_offset=lo, stats=st), want="keys")
    os.environ["PDA_SCORE_FUNNEL"] = "0"
    kr = ops.topk_merge(ops.score_topk_keys(U, Ish, users, K, ops.HEAD_RAW, None, hist, item_offset=lo), want="keys")
    torch.cuda.synchronize()
    same = torch.equal(kf, kr)
    ident = ops.kernel_identity(st["kernel_id"][0]) if "kernel_id" in st else {}
    nfb = int(st["fallback_rows"][0]) if "fallback_rows" in st else -1
    fb += max(nfb, 0)
    if not same or ident.get("geometry") != "funnel" or int(st["error"][0]) != 0:
        bad += 1
        print("CASE %d DIFFERS: d=%d items=%d (offset %d) users=%d K=%d bf16=%s scale=%g hist=%s rows differing %d ident %s err %d" %
              (c, d, Ish.shape[0], lo, nu, K, bf, scale, hist is not None, int((kf != kr).any(dim=1).sum()), ident, int(st["error"][0])), flush=True)
    elif c % 10 == 0:
        print("case %d ok: d=%d items=%d users=%d K=%d bf16=%s scale=%g hist=%s fallback rows %d" % (c, d, Ish.shape[0], nu, K, bf, scale, hist is not None, nfb), flush=True)
print("soak: %d cases, %d differing, %d rows through the fallback in all" % (n_cases, bad, fb))
