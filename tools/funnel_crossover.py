"""Funnel against generation 4 on random tables: where does the funnel start to pay?  tools/funnel_crossover.py [d=128]
(raw head, K = 50, train rows of 50 items; both forced with PDA_SCORE_FUNNEL = 1 / 0; the rule it produced: pda_score_topk_plan)"""
import os, sys, ctypes as C
import torch
sys.path.insert(0, ".")
from pda_amd import ops, _lib
d = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda")
L = _lib.load()
g = torch.Generator(device=dev).manual_seed(5)


def timeit(f, n=4):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for nI in ((4096, 8192, 16384, 32768) if os.environ.get("CROSS_SMALL") else ((20000, 200000) if os.environ.get("CROSS_FEW") else (16384, 32768, 65536, 200000))):
    I = torch.randn(nI, d, device=dev, generator=g) * 0.1
    for nu in ((64, 128, 256, 512, 1000) if os.environ.get("CROSS_FEW") else (1024, 2048, 4096, 16384, 65536)):
        U = torch.randn(nu, d, device=dev, generator=g) * 0.1
        users = torch.arange(nu, dtype=torch.int32, device=dev)
        cnt = torch.full((nu,), 50, dtype=torch.int64, device=dev)
        indptr = torch.zeros(nu + 1, dtype=torch.int64, device=dev); indptr[1:] = torch.cumsum(cnt, 0)
        idx = torch.sort(torch.randint(0, nI, (nu, 50), device=dev, generator=g, dtype=torch.int32), dim=1).values.reshape(-1).contiguous()
        hist = ops.HistoryCSR(indptr, idx, by_user=True)
        res = {}
        for name, env in (("gen4", "0"), ("funnel", "1")):
            os.environ["PDA_SCORE_FUNNEL"] = env
            st = {}
            f = lambda: ops.topk_merge(ops.score_topk_keys(U, I, users, 50, ops.HEAD_RAW, None, hist, stats=st), want="keys")
            keys = f()
            res[name] = (timeit(f), keys, int(st["fallback_rows"][0]) if "fallback_rows" in st else -1)
        same = all(torch.equal(res["gen4"][1], v[1]) for v in res.values())
        print("d=%d items %6d users %6d: " % (d, nI, nu) + "  ".join("%s %.3f ms%s" % (k, v[0], (" (%d fallback rows)" % v[2]) if v[2] > 0 else "") for k, v in res.items()) +
              ("  identical keys" if same else "  KEYS DIFFER"), flush=True)
