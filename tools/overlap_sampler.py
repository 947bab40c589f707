"""sample(t+1) || step(t): two batch buffers, the sampler on a side stream inside the captured graph."""
import sys, time, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic
dev = torch.device('cuda')
W = synthetic.make_workload('c2', dev)
B, G, regs, lr = 2048, 64, 1e-2, 1e-2
U, I = W.U.clone(), W.I.clone()
loss = torch.zeros(3, device=dev)
step_dev = torch.zeros(2, dtype=torch.int64, device=dev)
def mk():
    return (torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev),
            torch.empty(B, dtype=torch.float32, device=dev), torch.empty(B, dtype=torch.float32, device=dev))
bufs = [mk(), mk()]
def sample(into, parity):
    ops.sample_triplets_into(into, W.hist_indptr, W.hist_indices, seed=7, step_dev=step_dev, n_pool=W.n_users,
                             train_slots=W.hist_slots, neg_range=(0, W.n_items), pop_matrix=W.pop_train, parity=parity)
def step(b):
    ops.bpr_step(U, I, *b, regs=regs, reg_div=B, lr=lr, mode=ops.UPD_SGD_FUSED, loss_acc=loss)
def run(overlap):
    side = torch.cuda.Stream()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        sample(bufs[0], 0); sample(bufs[1], 1); step(bufs[0])
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        sample(bufs[0], 0)
        for i in range(G):
            if overlap:
                side.wait_stream(main)                   # step(i-1) has read bufs[(i+1)&1]; sample(i) has been enqueued
                with torch.cuda.stream(side):
                    sample(bufs[(i + 1) & 1], (i + 1) & 1)
                step(bufs[i & 1])
                main.wait_stream(side)
            else:
                step(bufs[i & 1])
                sample(bufs[(i + 1) & 1], (i + 1) & 1)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 16
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("overlap=%s: %.1f us per step, %.1f M triplets/s" % (overlap, dt / (reps * G) * 1e6, reps * G * B / dt / 1e6))
def run_fused():
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    kw = dict(n_pool=W.n_users, train_slots=W.hist_slots, neg_range=(0, W.n_items), pop_matrix=W.pop_train)
    def body(i):
        ops.bpr_step_and_sample(U, I, *bufs[i & 1], regs=regs, reg_div=B, lr=lr, next_out=bufs[(i + 1) & 1], train_indptr=W.hist_indptr,
                                train_indices=W.hist_indices, seed=7, step_dev=step_dev, parity=(i + 1) & 1, loss_acc=loss, **kw)
    with torch.cuda.stream(s):
        sample(bufs[0], 0); body(0); body(1)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(G): body(i)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 16
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("one launch: %.1f us per step, %.1f M triplets/s" % (dt / (reps * G) * 1e6, reps * G * B / dt / 1e6))
run(False); run(True); run_fused()
