"""SGD step vs batch size: where the step stops being launch/latency-bound, and what the exact planned step (no atomics) does there.
Per B, HIP graphs of 64 launches over 16 pre-staged batches: (a) the fused hogwild step (pda_bpr_step_f32 PDA_UPD_SGD_FUSED, user rows
stored plainly while users are distinct: B <= n_users), (b) the exact planned step (pda_bpr_step_plan_f32; plans made beforehand:
pda_triplet_plan up to 4 096 triplets, pda_triplet_plan_large beyond), and the time of one plan.
usage: b_sweep.py [workload=c2] [B ...] >> profiles/roundN_train_b_sweep.txt"""
import sys, time, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic
dev = torch.device("cuda")
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
Bs = [int(x) for x in sys.argv[2:]] or [2048, 4096, 8192, 16384, 32768, 65536]
W = synthetic.make_workload(wl, dev)
regs, lr = 1e-2, 1e-2
bpt = 6 * W.d * 4 + 20
print("# %s (%d users x %d items, d = %d), PD/PDA loss; algorithmic bytes per triplet = 6 rows x %d B + 20 = %d; HBM frac = of 8 TB/s"
      % (W.name.upper(), W.n_users, W.n_items, W.d, W.d * 4, bpt))
print("# %8s | %s | %s | %s" % ("B", "fused (hogwild): us/step  M triplets/s  HBM frac", "exact planned: us/step  M triplets/s  HBM frac", "one plan: us"))


def graph_time(body, B):
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
    return (time.perf_counter() - t0) / (reps * 64) * 1e6


for B in Bs:
    distinct = B <= W.n_users
    U, I, loss = W.U.clone(), W.I.clone(), torch.zeros(3, device=dev)
    grouped = B <= 4096
    batches = [ops.sample_triplets(W.hist_indptr, W.hist_indices, B, seed=2020, step=s, n_pool=W.n_users, train_slots=W.hist_slots,
                                   neg_range=(0, W.n_items), pop_matrix=W.pop_train, sort_by_pos=grouped) for s in range(16)]
    us_f = graph_time(lambda i: ops.bpr_step(U, I, *batches[i % 16], regs=regs, reg_div=B, lr=lr, mode=ops.UPD_SGD_FUSED, loss_acc=loss,
                                             grouped=grouped, users_distinct=distinct), B)
    fused = "%24.2f %13.1f %9.3f" % (us_f, B / us_f, B / us_f * 1e6 * bpt / 1e9 / 8000.0)
    if distinct:
        U, I = W.U.clone(), W.I.clone()
        plans = [ops.triplet_plan(*b[:3])[0] for b in batches]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in batches: ops.triplet_plan(*b[:3])
        torch.cuda.synchronize()
        us_plan = (time.perf_counter() - t0) / 16 * 1e6
        assert ops.plan_header(plans[0])[1] == 0
        scratch = [None]
        def body(i):
            scratch[0] = ops.bpr_step_plan(U, I, *batches[i % 16], regs=regs, reg_div=B, lr=lr, plan=plans[i % 16], scratch=scratch[0], loss_acc=loss)
        us_p = graph_time(body, B)
        planned = "%22.2f %13.1f %9.3f" % (us_p, B / us_p, B / us_p * 1e6 * bpt / 1e9 / 8000.0)
        plan_s = "%10.1f" % us_plan
    else:
        planned, plan_s = "%46s" % "(B > n_users: users repeat, no exact plan)", "%10s" % "-"
    print("  %8d | %s | %s | %s" % (B, fused, planned, plan_s), flush=True)
