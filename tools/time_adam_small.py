#!/usr/bin/env python
"""The reference-faithful Adam step on the small tables (C1 / C2): round 5's five launches against round 6's two (pda_adam_step_f32), each cache
policy.  usage: python tools/time_adam_small.py [c2|c1] [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv, argv = sys.argv[:1], sys.argv[1:]
import bench  # noqa: E402
from pda_amd import ops, synthetic  # noqa: E402

wl = argv[0] if argv else "c2"
steps = int(argv[1]) if len(argv) > 1 else 1024
dev = torch.device("cuda", 0)
W = synthetic.make_workload(wl, dev)
B, regs, lr, NB = 2048, 1e-2, 1e-2, 64
batches = [ops.sample_triplets(W.hist_indptr, W.hist_indices, B, seed=2020, step=s, n_pool=W.n_users, train_slots=W.hist_slots,
                               neg_range=(0, W.n_items), pop_matrix=W.pop_train, sort_by_pos=True) for s in range(NB)]
loss = torch.zeros(3, device=dev)
sweep_bytes = 6 * (W.n_users + W.n_items) * W.d * 4


def report(name, r):
    print("%-44s %8.2f us/step  %6.2f TB/s of the sweep's algorithmic bytes (%.3f of 8 TB/s)" %
          (name, r["us_per_step"], sweep_bytes / r["us_per_step"] / 1e6, sweep_bytes / r["us_per_step"] / 1e6 / 8), flush=True)


def fresh():
    U, I = W.U.clone(), W.I.clone()
    return U, I, [torch.zeros_like(t) for t in (U, U, U, I, I, I)]


U, I, st = fresh()
tb = ops.adam_touched_bitmaps(W.n_users, W.n_items, dev)
tc = [0]


def five(i):
    tc[0] += 1
    bt = batches[i % NB]
    ops.bpr_step(U, I, *bt, regs=regs, reg_div=B, mode=ops.UPD_DENSE_GRAD, gU=st[2], gI=st[5], loss_acc=loss)
    ops.adam_mark_rows(bt[0], bt[1], bt[2], tb[0], tb[1])
    ops.adam_dense_sweep3(U, st[0], st[1], st[2], tb[0], I, st[3], st[4], st[5], tb[1], ops.adam_lr_t(lr, min(tc[0], 1000)))


report("five launches (round 5)", bench.timed_graph_steps(five, steps, B, 64))
for pol, nm in ((1, "resident"), (2, "streaming"), (0, "auto")):
    for distinct in (True, False):
        U, I, st = fresh()
        tags = ops.adam_row_tags(W.n_users, W.n_items, dev)
        tc[0] = 0

        def two(i):
            tc[0] += 1
            ops.adam_step(U, st[0], st[1], st[2], tags[0], I, st[3], st[4], st[5], tags[1], *batches[i % NB], regs=regs, reg_div=B, step=tc[0],
                          lr_t=ops.adam_lr_t(lr, min(tc[0], 1000)), grouped=True, users_distinct=distinct, cache_policy=pol, loss_acc=loss)
        report("two launches, %s, users_distinct=%d" % (nm, distinct), bench.timed_graph_steps(two, steps, B, 64))
# the parts alone
U, I, st = fresh()
tags = ops.adam_row_tags(W.n_users, W.n_items, dev)
for pol, nm in ((1, "resident"), (2, "streaming")):
    report("sweep4 alone, %s" % nm, bench.timed_graph_steps(
        lambda i: ops.adam_dense_sweep4(U, st[0], st[1], st[2], tags[0], I, st[3], st[4], st[5], tags[1], 1, 1e-3, cache_policy=pol), steps, B, 64))
report("bpr_step(DENSE_GRAD) alone", bench.timed_graph_steps(
    lambda i: ops.bpr_step(U, I, *batches[i % NB], regs=regs, reg_div=B, mode=ops.UPD_DENSE_GRAD, gU=st[2], gI=st[5], loss_acc=loss, grouped=True), steps, B, 64))
