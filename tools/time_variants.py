"""Time A/B builds of the library (tools/build_variant.sh) in ONE process on a reduced bench workload.
usage: time_variants.py tag1,tag2,... [mode=order|0|1] [head=1] [users per block=65536] [workload=c3] [n_users=131072]
tag 'base' = the regular libpda_hip.so"""
import os, sys, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic, _lib
tags = sys.argv[1].split(",")
mode = {"order": "order", "0": False, "1": True}[sys.argv[2] if len(sys.argv) > 2 else "order"]
head = int(sys.argv[3]) if len(sys.argv) > 3 else 1
Bu = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
wl = sys.argv[5] if len(sys.argv) > 5 else "c3"
nus = int(sys.argv[6]) if len(sys.argv) > 6 else 131072
dev = torch.device('cuda')
W = synthetic.make_workload(wl, dev, n_users=nus)
hist = None if os.environ.get('NOHIST') else ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
blocks = [torch.arange(s, s + Bu, dtype=torch.int32, device=dev) for s in range(0, W.n_users - Bu + 1, Bu)][:4]
os.environ["PDA_SCORE_KERNEL"] = os.environ.get("PDA_SCORE_KERNEL", "v4")
here = os.path.dirname(os.path.abspath(_lib.__file__))
for rep in range(2):
    for tag in tags:
        _lib._lib = None
        _lib.LIB_PATH = os.path.join(here, "csrc", "libpda_hip.so" if tag == "base" else "variants/libpda_hip_%s.so" % tag)
        pop = W.pop_last if head else None
        st = {}
        ops.score_topk_keys(W.U, W.I, blocks[0], 50, head, pop, hist, prune=mode, stats=st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 3
        for i in range(n):
            for b in blocks:
                ops.score_topk_keys(W.U, W.I, b, 50, head, pop, hist, prune=mode, stats=st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (n * len(blocks))
        fl = 2.0 * Bu * W.n_items * W.d / (ms * 1e-3) / 1e12
        print("%-10s %.3f ms  %.2f M users/s  %.0f TF (%.3f)  err=%d cand/user=%.0f" % (tag, ms, Bu / ms / 1e3, fl, fl / 2500,
              int(st["pairs_rescored"].view(torch.int32)[0].item() * 0 + torch.zeros(1).item()), float(st["pairs_rescored"][0]) / Bu), flush=True)
