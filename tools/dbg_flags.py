import ctypes as C, sys, torch, numpy as np
sys.path.insert(0, '.')
from pda_amd import ops, synthetic, _lib
from pda_amd._lib import ptr, stream_ptr
dev = torch.device('cuda')
W = synthetic.make_workload('c3', dev)
lib = _lib.load()
hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
nu = 2048
users = torch.arange(0, nu, dtype=torch.int32, device=dev)
prep = ops.item_prep(W.I)
n, d = W.I.shape
ws = torch.zeros(lib.pda_score_topk_workspace_bytes(nu), dtype=torch.uint8, device=dev)
out = torch.empty((1, nu, 50), dtype=torch.int64, device=dev)
for head, pop in ((1, W.pop_last), (0, None)):
    rc = lib.pda_score_topk_prepped_f32(ptr(W.U), ptr(W.I), ptr(prep), ptr(pop), ptr(users), nu, 0, n, d, ptr(hist.indptr), ptr(hist.indices), 1, 50, head, 1, ptr(out), ptr(ws), stream_ptr())
    torch.cuda.synchronize()
    w = ws.cpu().numpy()
    popmax = w[:4].view(np.float32)[0]
    flags = w[16:16 + 4 * (nu // 128)].view(np.int32)
    plane = ((n * d * 2 + 255) // 256) * 256
    nrm_off = 2 * plane
    nb = ((n * 4 + 255) // 256) * 256
    pb = prep.cpu().numpy()
    nimax = pb[nrm_off + nb: nrm_off + nb + 4].view(np.float32)[0]
    norms = pb[nrm_off: nrm_off + 4 * n].view(np.float32)
    print("head", head, "rc", rc, "popmax", popmax, "NI_MAX", nimax, "true max norm", norms.max(), "flagged tiles", int(flags.sum()), "of", len(flags))
