import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import time_emit as T
from pda_amd import ops
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(7)
d = 128
nU, nI, n_users = 5000, 9000, 2048
U = (torch.randn(nU, d, generator=g) * 0.1).to(dev)
I = (torch.randn(nI, d, generator=g) * 0.1).to(dev)
users = torch.arange(n_users, dtype=torch.int32, device=dev)
prep = ops.item_prep4(I, None, None)
cap, S = 256, 1
tot, offs = T.layout(n_users, d, S, cap)
ws = torch.zeros(tot, dtype=torch.uint8, device=dev)
thr = torch.full((n_users,), 1e30, device=dev)
T.run(U, users, prep, nI, d, thr, 0, 10**6, S, cap, ws)
torch.cuda.synchronize()
cnt = T.counts(ws, offs, n_users, d, S).cpu()
print("counts wave0 u0:", cnt[0, 0, 0, 0].tolist())
print("counts wave0 u8:", cnt[0, 0, 0, 8].tolist())
print("total", int(cnt.sum()), "half-tiles", 2 * ((nI + 63) // 64))
es, ls = 64 * 16 * 48, 16 * 48
wsc = ws.cpu()
for (u, lane) in ((0, 0), (8, 5)):
    c = int(cnt[0, 0, 0, u, lane])
    for e in list(range(min(c, 6))):
        o = offs[3] + e * es + lane * ls + u * 48
        w = wsc[o:o + 48].view(torch.float32)
        h = int(wsc[o + 32:o + 36].view(torch.int32)[0])
        print(u, lane, e, "h", h, [float(x) for x in w[:8]])
PL = T.prep_layout(nI, d)
meta = prep[PL["meta5"]:PL["meta5"] + PL["n_tiles"] * 2 * 16].view(torch.float32).view(-1, 4).cpu()
print("meta5 rows 0..11:", meta[:12].tolist())
r5 = prep[PL["rows5"]:PL["rows5"] + PL["n_tiles"] * 2 * 64 * d].view(torch.bfloat16).view(-1, 32, d).float().cpu()   # [half-tile][row][swizzled chunks]
Ib = I.bfloat16().float().cpu()
for ht in (0, 1, 2, 3, 4, 5, 8):
    err = 0.0
    for r in range(32):
        sw = (r & 15) if d >= 128 else ((r >> 1) & 7)
        row = torch.cat([r5[ht, r, 8 * (e ^ sw):8 * (e ^ sw) + 8] for e in range(d // 8)])
        err = max(err, float((row - Ib[ht * 32 + r]).abs().max()))
    print("half-tile", ht, "max |image - bf16(I)| =", err)
import os
if os.environ.get("DUMP"):
    hi = int(os.environ["DUMP"])
    ws.zero_()
    T.run(U, users, prep, nI, d, thr, 0, hi, S, cap, ws)
    torch.cuda.synchronize()
    utiles = 2
    base = offs[3] + utiles * S * 4 * cap * es
    lds = ws[base:base + 8 * 8448].cpu()
    img = prep[PL["rows5"]:PL["rows5"] + PL["n_tiles"] * 2 * 64 * d].cpu()
    for slot in range(8):
        sl = lds[slot * 8448:slot * 8448 + 8192]
        match = [ht for ht in range(0, 2 * hi + 2) if ht < PL["n_tiles"] * 2 and torch.equal(sl, img[ht * 8192:(ht + 1) * 8192])]
        nz = int((sl != 0).sum())
        m4 = lds[slot * 8448 + 8192:slot * 8448 + 8208].view(torch.float32).tolist()
        print("LDS slot", slot, "equals half-tile", match, "nonzero bytes", nz, "meta", m4)
