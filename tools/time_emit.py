"""The funnel's emitting sweep (sweep7_kernel) alone, through the debug entry pda_debug_emit_sweep: what does the loop cost at a given rate of
entries per user?  And (mode check) are the entries the ones torch finds with the same fp16 products?

usage: time_emit.py check [d=128]                         small case against torch
       time_emit.py time [workload=c3] [users=262144]     thresholds at several sample ranks -> entries per user, ms, fraction of 2.5 PF
"""
import ctypes as C
import os
import sys

import numpy as np

import torch

sys.path.insert(0, ".")
from pda_amd import _lib, ops, synthetic  # noqa: E402

_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t


def lib():
    L = _lib.load()
    L.pda_debug_emit_layout.restype, L.pda_debug_emit_layout.argtypes = _sz, [_i, _i, _i, _i, C.POINTER(_sz)]
    L.pda_debug_emit_sweep.restype, L.pda_debug_emit_sweep.argtypes = _i, [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp]
    return L


def layout(n_users, d, S, cap):
    offs = (_sz * 5)()
    tot = lib().pda_debug_emit_layout(n_users, d, S, cap, offs)
    return tot, list(offs)


def prep_layout(n, d):
    al = lambda x: (x + 255) & ~255
    nt = (n + 63) // 64
    pos_of = 256
    sufA = pos_of + al(n * 4)
    sufB = sufA + al(nt * 4)
    sufR = sufB + al(nt * 4)
    rows = sufR + al(nt * 4)
    total = rows + nt * 64 * (2 * d + 48)
    rows5 = al(total)
    meta5 = rows5 + al(nt * 2 * 64 * d)
    return dict(n_tiles=nt, rows5=rows5, meta5=meta5)


def run(U, users, prep, n_items, d, thr, lo, hi, S, cap, ws):
    ptr = _lib.ptr
    _lib.check(lib().pda_debug_emit_sweep(ptr(U), int(U.dtype == torch.bfloat16), ptr(users), users.numel(), ptr(prep), n_items, d, ptr(thr), lo, hi, S, cap,
                                          ptr(ws), _lib.stream_ptr()), "pda_debug_emit_sweep")


def counts(ws, offs, n_users, d, S):
    ut = 512 if d == 256 else 1024
    nu = ut // 64
    es = 64 * nu * 48
    utiles = -(-n_users // ut)
    cur = ws[offs[2]:offs[2] + utiles * S * 4 * nu * 64 * 4].view(torch.int32).view(utiles, S, 4, nu, 64)
    return cur // es          # [utile][split][wave][u][lane]


def check(d):
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(7)
    nU, nI, n_users = 5000, 9000, 3000
    U = (torch.randn(nU, d, generator=g) * 0.1).to(dev)
    I = (torch.randn(nI, d, generator=g) * 0.1 * (0.5 + torch.rand(nI, 1, generator=g))).to(dev)
    users = torch.randperm(nU, generator=g)[:n_users].to(torch.int32).to(dev)
    order = ops.visiting_order(I, None)
    prep = ops.item_prep7(I, order)
    PL = prep_layout(nI, d)
    ut = 512 if d == 256 else 1024
    nu = ut // 64
    es, ls = 64 * nu * 48, nu * 48
    for S, lo, hi, q in ((1, 0, 10 ** 6, 0.995), (3, 2, 40, 0.98), (1, 0, 4, -1.0)):
        cap = 64
        tot, offs = layout(n_users, d, S, cap)
        ws = torch.zeros(tot, dtype=torch.uint8, device=dev)
        Ub, Ib = U[users.long()].half().float(), I.half().float()
        s = Ub @ Ib[order.long()].T                        # [n_users, pos]
        thr = torch.quantile(s[:, :2000], q, dim=1).contiguous() if q > 0 else torch.full((n_users,), -float("inf"), device=dev)
        run(U, users, prep, nI, d, thr, lo, hi, S, cap, ws)
        torch.cuda.synchronize()
        cnt = counts(ws, offs, n_users, d, S).cpu()
        eu = ws[offs[1]:offs[1] + (-(-n_users // ut)) * 32].view(torch.float32).cpu()
        meta = prep[PL["meta5"]:PL["meta5"] + PL["n_tiles"] * 2 * 16].view(torch.float32).view(-1, 4).cpu()
        wsc = ws.cpu()
        s = s.cpu()
        thr_c = thr.cpu()
        nt = PL["n_tiles"]
        bad = n_ent = n_exp = 0
        examples = []
        for rb in list(range(0, n_users, 97)) + [n_users - 1]:
            utile, wave, u, j = rb // ut, (rb % ut) // (ut // 4), (rb % (ut // 4)) // 16, rb % 16
            e_a, e_b = np.float32(eu[2 * (utile * 4 + wave)]), np.float32(eu[2 * (utile * 4 + wave) + 1])
            got = {}
            for sp in range(S):
                nts = (nt - sp + S - 1) // S if sp < nt else 0
                i0, i1 = min(nts, max(0, -(-(lo - sp) // S))), min(nts, max(0, -(-(hi - sp) // S)))
                widx = (utile * S + sp) * 4 + wave
                for hh in range(4):
                    lane = j + 16 * hh
                    c = int(cnt[utile, sp, wave, u, lane])
                    assert c <= cap
                    for e in range(c):
                        o = offs[3] + widx * cap * es + e * es + lane * ls + u * 48
                        w = wsc[o:o + 48].view(torch.float32)
                        h = int(wsc[o + 32:o + 36].view(torch.int32)[0])
                        if h >= 2 * (i1 - i0):
                            continue                      # written by the two half-tiles behind the end
                        T = sp + (i0 + (h >> 1)) * S
                        for r in range(8):
                            pos = T * 64 + (h & 1) * 32 + 16 * (r >> 2) + 4 * hh + (r & 3)
                            got[pos] = float(w[r])
                        n_ent += 1
            # expected: every (half-tile, quarter) whose maximum of s~ + ct beats the threshold
            for sp in range(S):
                nts = (nt - sp + S - 1) // S if sp < nt else 0
                for i in range(min(nts, max(0, -(-(lo - sp) // S))), min(nts, max(0, -(-(hi - sp) // S)))):
                    T = sp + i * S
                    for half in range(2):
                        mrow = meta[2 * T + half].numpy()
                        ct = float(np.float32(e_b * mrow[2]) + np.float32(np.float32(e_a * mrow[1]) + mrow[0]))      # (products of these magnitudes: fma = mul + add to 1e-9)
                        for hh in range(4):
                            ps = [T * 64 + half * 32 + 16 * ib + 4 * hh + r for ib in range(2) for r in range(4)]
                            vals = [(float(s[rb, p]) if p < nI else 0.0) + ct for p in ps]
                            m = max(vals)
                            t = float(thr_c[rb])
                            if m > t + 1e-5 * abs(t):
                                n_exp += 1
                                for p, v in zip(ps, vals):
                                    if p not in got or abs(got[p] - v) > 2e-5:
                                        bad += 1
                                        if len(examples) < 12:
                                            examples.append(("row", rb, "tile", T, "half", half, "hh", hh, "pos", p, "got", got.get(p), "want", v, "ct", ct, "thr", t))
                            elif m < t - 1e-5 * abs(t):
                                bad += sum(1 for p in ps if p in got)
        print("d=%d S=%d tiles [%d, %d) q=%.3f: %d entries read, %d expected groups, mismatches %d, stats error %d, kernel id %#x" %
              (d, S, lo, hi, q, n_ent, n_exp, bad, int(ws[0:4].view(torch.int32)[0]), int(ws[16:20].view(torch.int32)[0])))
        for e in examples:
            print("   ", e)
        assert bad == 0 or os.environ.get("EMIT_CHECK_GO_ON")


def time_(wl, Bu):
    dev = torch.device("cuda")
    W = synthetic.make_workload(wl, dev)
    d = W.d
    Bu = min(Bu, W.n_users)
    users = torch.arange(Bu, dtype=torch.int32, device=dev)
    order = ops.visiting_order(W.I, None)
    prep = ops.item_prep7(W.I, order)
    cap, S = 64, 1
    tot, offs = layout(Bu, d, S, cap)
    print("workspace %.2f GB" % (tot / 1e9))
    ws = torch.zeros(tot, dtype=torch.uint8, device=dev)
    m = 8192
    samp = torch.randperm(W.n_items, device=dev)[:m]
    Is = W.I[samp].half().float()
    tops = []
    for s0 in range(0, Bu, 16384):
        sc = W.U[s0:s0 + 16384].half().float() @ Is.T
        tops.append(torch.topk(sc, 64, dim=1).values)
    tops = torch.cat(tops)                                # [Bu, 64] descending
    for rank in (-1, 1, 2, 4, 8, 16, 32, 64):
        thr = torch.full((Bu,), 1e30, device=dev) if rank < 0 else tops[:, rank - 1].contiguous()
        run(W.U, users, prep, W.n_items, d, thr, 0, 10 ** 6, S, cap, ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 3
        e0.record()
        for _ in range(n):
            run(W.U, users, prep, W.n_items, d, thr, 0, 10 ** 6, S, cap, ws)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        cnt = counts(ws, offs, Bu, d, S)
        per_user = float(cnt.sum()) / Bu
        over = float((cnt > cap).float().mean())
        fl = 2.0 * Bu * W.n_items * d / (ms * 1e-3) / 1e12
        print("%s %d users, threshold = rank %d of a %d-item sample: %.1f entries per user (lists over capacity %.4f), %.3f ms, %.0f TF = %.3f of 2.5 PF" %
              (wl, Bu, rank, m, per_user, over, ms, fl, fl / 2500), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    if mode == "check":
        for d in ([int(sys.argv[2])] if len(sys.argv) > 2 else [128, 64, 256]):
            check(d)
    else:
        time_(sys.argv[2] if len(sys.argv) > 2 else "c3", int(sys.argv[3]) if len(sys.argv) > 3 else 262144)
