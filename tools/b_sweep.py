"""Fused SGD step vs batch size on BASELINE config 2 (50 000 x 20 000, d = 64): where the step stops being launch/latency-bound.
Per B: (a) one launch per step from a HIP graph of 64 launches, batches pre-staged; (b) pda_bpr_train_steps_f32, 64 steps per
launch, fresh batch every step.  usage: b_sweep.py > profiles/roundN_train_b_sweep.txt"""
import sys, time, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic
dev = torch.device("cuda")
W = synthetic.make_workload("c2", dev)
regs, lr = 1e-2, 1e-2
print("# %s, PD/PDA loss, fused SGD (pda_bpr_step_f32 PDA_UPD_SGD_FUSED); bytes per triplet = 6 rows x %d B + 20 = %d" % (W.name.upper(), W.d * 4, 6 * W.d * 4 + 20))
print("# %8s %22s %14s %10s   %22s %14s" % ("B", "graph: us/step", "M triplets/s", "HBM frac", "one launch: us/step", "M triplets/s"))
kw = dict(n_pool=W.n_users, train_slots=W.hist_slots, neg_range=(0, W.n_items), pop_matrix=W.pop_train)
for B in (2048, 4096, 8192, 16384, 32768, 65536):
    U, I, loss = W.U.clone(), W.I.clone(), torch.zeros(3, device=dev)
    batches = [ops.sample_triplets(W.hist_indptr, W.hist_indices, B, seed=2020, step=s, n_pool=W.n_users, train_slots=W.hist_slots,
                                   neg_range=(0, W.n_items), pop_matrix=W.pop_train, sort_by_pos=(B <= 4096)) for s in range(16)]
    body = lambda i: ops.bpr_step(U, I, *batches[i % 16], regs=regs, reg_div=B, lr=lr, mode=ops.UPD_SGD_FUSED, loss_acc=loss, grouped=(B <= 4096))
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(3): body(i)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(64): body(i)
    g.replay(); torch.cuda.synchronize()
    reps = max(2, 8192 // B)
    t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    us_g = (time.perf_counter() - t0) / (reps * 64) * 1e6
    # the device loop
    U, I = W.U.clone(), W.I.clone()
    mk = lambda: (torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev),
                  torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev))
    bufs = [mk(), mk()]
    ctr = torch.tensor([1], dtype=torch.int64, device=dev)
    ops.sample_triplets_into(bufs[0], W.hist_indptr, W.hist_indices, seed=7, step_dev=ctr, **kw)
    ws = torch.zeros(2, dtype=torch.int32, device=dev)
    go = lambda: ops.bpr_train_steps(U, I, bufs, 64, regs=regs, reg_div=B, lr=lr, train_indptr=W.hist_indptr, train_indices=W.hist_indices,
                                     seed=7, step_ctr=ctr, loss_acc=loss, barrier_ws=ws, **kw)
    go(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): go()
    torch.cuda.synchronize()
    us_l = (time.perf_counter() - t0) / (reps * 64) * 1e6
    assert int(ws[1]) == 0
    bpt = 6 * W.d * 4 + 20
    print("  %8d %22.2f %14.1f %10.3f   %22.2f %14.1f" % (B, us_g, B / us_g, B / us_g * 1e6 * bpt / 1e9 / 8000.0, us_l, B / us_l))
