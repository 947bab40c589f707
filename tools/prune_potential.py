import sys, torch
sys.path.insert(0, '.')
from pda_amd import ops, synthetic
dev = torch.device('cuda')
W = synthetic.make_workload('c3', dev)
hist = ops.HistoryCSR(W.hist_indptr, W.hist_indices, by_user=True)
users = torch.arange(0, 16384, dtype=torch.int32, device=dev)
idx, val = ops.recommend_topk(W.U, W.I, users, 50, 1, W.pop_last, hist)
tau = val[:, -1]
un = W.U[:16384].norm(dim=1)
nimax = W.I.norm(dim=1).max()
gate_row = tau / (1 + un * nimax)
gate_wg = gate_row.view(-1, 128).min(dim=1).values
pop = W.pop_last
nt = (pop.numel() + 31) // 32
tp = torch.nn.functional.pad(pop, (0, nt * 32 - pop.numel())).view(nt, 32).max(dim=1).values
for g in (gate_wg.min(), gate_wg.median(), gate_wg.max()):
    print("gate %.4f -> tiles that cannot be skipped at the END: %d of %d" % (float(g), int((tp > g).sum()), nt))
print("tau: min %.3f med %.3f | pop quantiles:" % (float(tau.min()), float(tau.median())), [round(float(pop.quantile(q)), 4) for q in (0.5, 0.9, 0.99, 0.999)], "max", float(pop.max()))
# tighter bound: per-item norm instead of NI_MAX
