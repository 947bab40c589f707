import os, sys, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic
dev = torch.device('cuda')
W = synthetic.make_workload("c3", dev)
hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
def run(Bu, lists, ns, nloc=None):
    os.environ["PDA_SCORE_LISTS"] = lists
    I = W.I if nloc is None else W.I[:nloc].contiguous()
    pop = W.pop_last if nloc is None else W.pop_last[:nloc].contiguous()
    users = torch.arange(Bu, dtype=torch.int32, device=dev)
    f = lambda: ops.score_topk_keys(W.U, I, users, 50, ops.HEAD_POP, pop, hist, prune="order", n_splits=ns)
    k = f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): k = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5, ops.topk_merge(k, want="keys")
for Bu, nloc in ((65536, None), (65536, 100000), (98304, None), (131072, 50000)):
    ref = None
    for lists, ns in (("lds", 0), ("wide", 1), ("wide", 2), ("wide", 4)):
        ms, keys = run(Bu, lists, ns, nloc)
        same = "" if ref is None else " same=%s" % torch.equal(ref, keys)
        ref = keys if ref is None else ref
        print("users %6d items %6s  %-5s splits %d: %.3f ms%s" % (Bu, nloc or W.n_items, lists, ns, ms, same), flush=True)
