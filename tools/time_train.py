"""Step-time comparison of the SGD paths on one GPU (HIP graphs of 64 steps over 64 pre-staged device-sampled batches):
fused hogwild step (grouped batches / sampling order), the old exact pair (PDA_UPD_NONE + pda_sgd_apply_f32), the planned exact
step (two launches, no atomics) and the planned one-launch step.   python tools/time_train.py [c2|c3|c5shard] [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pda_amd import ops, synthetic

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
Bs = [int(x) for x in sys.argv[2:]] or [2048, 4096]
dev = torch.device("cuda")
W = synthetic.make_workload(wl, dev)
bf = wl == "c5shard"
regs, lr, NB, G = 1e-2, 1e-2, 64, 64
print("workload %s: %d x %d, d = %d%s" % (wl, W.n_users, W.n_items, W.d, " (bf16 shadows + fp32 masters)" if bf else ""))


def timed_graph(body, n_steps=2048):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(3):
            body(i)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(G):
            body(i)
    reps = max(1, n_steps // G)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * G) * 1e6


ctr = torch.zeros(1, dtype=torch.int64, device=dev)
from pda_amd import _lib
floor = timed_graph(lambda i: _lib.check(_lib.load().pda_counter_add(_lib.ptr(ctr), 1, _lib.stream_ptr()), "pda_counter_add"))
print("launch floor: a one-thread kernel (pda_counter_add) as a node of the same 64-node graphs: %.2f us per node" % floor)

for B in Bs:
    raw = [ops.sample_triplets(W.hist_indptr, W.hist_indices, B, seed=2020, step=s, n_pool=W.n_users, train_slots=W.hist_slots,
                               neg_range=(0, W.n_items), pop_matrix=W.pop_train) for s in range(NB)]
    grouped = [tuple(t.clone() for t in b) for b in raw]
    if B <= 4096:
        for b in grouped:
            ops.group_triplets_by_pos(*b)
    plans_raw = [ops.triplet_plan(b[0], b[1], b[2])[0] for b in raw] if B <= 4096 else None
    plans_grp = [ops.triplet_plan(b[0], b[1], b[2])[0] for b in grouped] if B <= 4096 else None
    loss = torch.zeros(3, device=dev)
    res = {}
    bytes_per = 6 * W.d * (2 if bf else 4) + 20
    if bf:
        U16, I16 = W.U.to(torch.bfloat16), W.I.to(torch.bfloat16)
        Um, Im = U16.float(), I16.float()
        res["fused + 3 refreshes (bf16)"] = timed_graph(lambda i: ops.bpr_step_bf16(U16, I16, *raw[i % NB], regs=regs, reg_div=B, lr=lr, mode=ops.UPD_SGD_FUSED,
                                                                                 U_master=Um, I_master=Im, loss_acc=loss))
        if plans_raw:
            sc = [None]
            def body(i):
                sc[0] = ops.bpr_step_plan(U16, I16, *raw[i % NB], regs=regs, reg_div=B, lr=lr, plan=plans_raw[i % NB], scratch=sc[0], loss_acc=loss, U_master=Um, I_master=Im)
            res["planned exact (bf16, 2 launches)"] = timed_graph(body)
    else:
        U, I = W.U.clone(), W.I.clone()
        res["fused, grouped batches"] = timed_graph(lambda i: ops.bpr_step(U, I, *grouped[i % NB], regs=regs, reg_div=B, lr=lr, mode=ops.UPD_SGD_FUSED, loss_acc=loss, grouped=B <= 4096))
        res["fused, sampling order"] = timed_graph(lambda i: ops.bpr_step(U, I, *raw[i % NB], regs=regs, reg_div=B, lr=lr, mode=ops.UPD_SGD_FUSED, loss_acc=loss))
        sc0 = [None]
        def old_exact(i):
            sc0[0] = ops.sgd_step_exact(U, I, *raw[i % NB], regs=regs, reg_div=B, lr=lr, loss_acc=loss, scratch=sc0[0])
        res["exact, PDA_UPD_NONE + pda_sgd_apply"] = timed_graph(old_exact)
        if plans_raw:
            sc = [None]
            def planned(i):
                sc[0] = ops.bpr_step_plan(U, I, *raw[i % NB], regs=regs, reg_div=B, lr=lr, plan=plans_raw[i % NB], scratch=sc[0], loss_acc=loss)
            res["planned exact (2 launches)"] = timed_graph(planned)
            res["planned one launch (sampling order)"] = timed_graph(lambda i: ops.bpr_step_plan(U, I, *raw[i % NB], regs=regs, reg_div=B, lr=lr, plan=plans_raw[i % NB], exact=False, loss_acc=loss))
            res["planned one launch (grouped)"] = timed_graph(lambda i: ops.bpr_step_plan(U, I, *grouped[i % NB], regs=regs, reg_div=B, lr=lr, plan=plans_grp[i % NB], exact=False, loss_acc=loss))
            stack = [torch.stack([b[k] for b in raw[:32]]) for k in range(3)]
            out = ops.triplet_plan(*stack)
            res["plan of 32 batches (one launch) / 32"] = timed_graph(lambda i: ops.triplet_plan(*stack, out=out), 512) / 32
    for k, v in res.items():
        print("B = %6d  %-44s %8.2f us/step  %8.1f M triplets/s  %5.1f %% of HBM (%d B/triplet)" % (B, k, v, B / v, B / v * 1e6 * bytes_per / 8e12 * 100, bytes_per))
