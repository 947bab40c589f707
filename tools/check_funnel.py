"""The funnel against generation 4 on the same inputs: identical merged keys?  How many rows took the exact fallback, and how long does a call take?
usage: check_funnel.py small | c3 [users=262144] | c2"""
import os, sys, time
import torch
sys.path.insert(0, ".")
from pda_amd import ops, synthetic, _lib
import ctypes as C

dev = torch.device("cuda")


def both(U, I, users, K, hist):
    os.environ["PDA_SCORE_FUNNEL"] = "1"
    st = {}
    kf = ops.score_topk_keys(U, I, users, K, ops.HEAD_RAW, None, hist, stats=st)
    kf = ops.topk_merge(kf, want="keys")
    os.environ["PDA_SCORE_FUNNEL"] = "0"
    kr = ops.topk_merge(ops.score_topk_keys(U, I, users, K, ops.HEAD_RAW, None, hist), want="keys")
    torch.cuda.synchronize()
    bad = (kf != kr).any(dim=1)
    return kf, kr, int(bad.sum()), st


def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def sched(n_items, K):
    L = _lib.load()
    out = (C.c_int * 96)()
    L.pda_debug_funnel_schedule.restype = C.c_int
    n = L.pda_debug_funnel_schedule(n_items, K, out, 32)
    return [(out[3 * i], out[3 * i + 1], out[3 * i + 2]) for i in range(n)]


def diag(st, nu, nloc, d, kf, kr):
    L = _lib.load()
    offs = (C.c_size_t * 10)()
    L.pda_debug_funnel_layout.restype = C.c_int
    sc = L.pda_debug_funnel_layout(nu, nloc, d, offs)
    S, cap = sc >> 16, sc & 0xFFFF
    ws = st["workspace"]
    f32 = lambda o: ws[o:o + nu * 4].view(torch.float32)
    i32 = lambda o: ws[o:o + nu * 4].view(torch.int32)
    thr, tk, tmax, ncand, flags = f32(offs[0]), f32(offs[1]), f32(offs[2]), i32(offs[3]), i32(offs[4])
    hard = (flags & 1) != 0
    soft = (~hard) & ((flags & 2) != 0)
    print(" splits %d cap %d: hard %d soft %d; ncand mean %.1f max %d; tk mean %.4f tmax mean %.4f" %
          (S, cap, int(hard.sum()), int(soft.sum()), float(ncand.float().mean()), int(ncand.max()), float(tk[torch.isfinite(tk)].mean()), float(tmax[torch.isfinite(tmax)].mean())))
    ut = 1024
    per = [int(hard[i:i + ut * 16].sum()) for i in range(0, nu, ut * 16)]
    print(" hard rows per 16 user tiles:", per)
    bad = (kf != kr).any(dim=1)
    print(" differing rows: hard %d soft %d neither %d" % (int((bad & hard).sum()), int((bad & soft).sum()), int((bad & ~hard & ~soft).sum())))
    nu_ = ut // 64
    es = 64 * nu_ * 48
    utiles = -(-nu // ut)
    cur = ws[offs[8]:offs[8] + utiles * S * 4 * nu_ * 64 * 4].view(torch.int32).view(utiles, S, 4, nu_, 64) // es
    print(" last launch: entries per list mean %.2f max %d, lists over capacity %d" % (float(cur.float().mean()), int(cur.max()), int((cur > cap).sum())))


mode = sys.argv[1] if len(sys.argv) > 1 else "small"
if mode == "small":
    g = torch.Generator(device="cpu").manual_seed(11)
    for d in (128, 64):
        for (nU, nI, nu) in ((7000, 20000, 6000), (5000, 70000, 4097)):
            U = (torch.randn(nU, d, generator=g) * 0.1).to(dev)
            I = (torch.randn(nI, d, generator=g) * 0.1 * (0.5 + torch.rand(nI, 1, generator=g))).to(dev)
            users = torch.randperm(nU, generator=g)[:nu].to(torch.int32).to(dev)
            rows = [torch.randperm(nI, generator=g)[:int(torch.randint(0, 60, (1,), generator=g))].tolist() for _ in range(nU)]
            hist = ops.HistoryCSR.from_lists(rows, dev, by_user=True)
            for h in (hist, None):
                kf, kr, nbad, st = both(U, I, users, 50, h)
                print("d=%d %d users x %d items hist=%s: rows that differ %d, fallback rows %d, error %d, pairs rescored per user %.1f, kernel %s, schedule %s" %
                      (d, nu, nI, h is not None, nbad, int(st["fallback_rows"][0]), int(st["error"][0]), float(st["pairs_rescored"][0]) / nu,
                       ops.kernel_identity(st["kernel_id"][0]), sched(nI, 50)), flush=True)
                if nbad:
                    r = int((kf != kr).any(dim=1).nonzero()[0])
                    print(" row", r, "funnel", kf[r, :8].tolist(), "gen4", kr[r, :8].tolist())
else:
    W = synthetic.make_workload(mode, dev)
    Bu = min(int(sys.argv[2]) if len(sys.argv) > 2 else 262144, W.n_users)
    users = torch.arange(Bu, dtype=torch.int32, device=dev)
    hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
    print("schedule", sched(W.n_items, 50))
    kf, kr, nbad, st = both(W.U, W.I, users, 50, hist)
    print("%s %d users: rows that differ %d, fallback rows %d, error %d, pairs rescored per user %.1f, kernel %s" %
          (mode, Bu, nbad, int(st["fallback_rows"][0]), int(st["error"][0]), float(st["pairs_rescored"][0]) / Bu, ops.kernel_identity(st["kernel_id"][0])), flush=True)
    diag(st, Bu, W.n_items, W.d, kf, kr)
    fl = 2.0 * Bu * W.n_items * W.d
    os.environ["PDA_SCORE_FUNNEL"] = "1"
    ms = timeit(lambda: ops.score_topk_keys(W.U, W.I, users, 50, ops.HEAD_RAW, None, hist))
    print("funnel: %.3f ms = %.3f of 2.5 PF" % (ms, fl / (ms * 1e-3) / 2.5e15))
    os.environ["PDA_SCORE_FUNNEL"] = "0"
    ms = timeit(lambda: ops.score_topk_keys(W.U, W.I, users, 50, ops.HEAD_RAW, None, hist), n=2)
    print("generation 4: %.3f ms = %.3f of 2.5 PF" % (ms, fl / (ms * 1e-3) / 2.5e15))
