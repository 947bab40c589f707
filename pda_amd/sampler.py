"""Triplet samplers.

Sampler protocol of the reference (MF/train_new_api.py:178-220, 260-288, 366-412): a zero-argument generator
yielding, per step, a tuple of sequences of length batch_size -- (users, pos, neg) for BPRMF or
(users, pos, neg, pos_pop, neg_pop) for PD/PDA -- exactly `n_train // batch_size + 1` times per epoch (:190).

    host_generator   single-process restatement of the reference's Python generators (same distribution:
                     rd.sample users, uniform positive with its time slot, rejection-sampled negative);
                     yields python lists like the reference, for drop-in use and for injecting fixed batches.
    DeviceSampler    pda_sample_triplets (HIP): counter-based, O(B) per batch, tensors never leave HBM.
                     Not stream-identical to the host generators -- parity is defined on injected batches
                     (last bullet).
"""
from __future__ import annotations

import random as rd

import numpy as np
import torch

from . import ops


def n_batches(data) -> int:
    return data.n_train // data.batch_size + 1


def host_generator(data, with_pop: bool):
    """generator_n_batch (:260-288) / generator_n_batch_with_pop (:366-412), one process, one epoch."""
    all_users = list(data.train_user_list.keys())
    bs = data.batch_size
    for _ in range(n_batches(data)):
        if bs <= data.n_users:
            users = rd.sample(all_users, bs)                    # :380-381 unique users
        else:
            users = [rd.choice(all_users) for _ in range(bs)]   # :383
        pos, neg, ppop, npop = [], [], [], []
        for u in users:
            clicked = data.train_user_list[u]
            if not clicked:                                     # :387-390
                p, t = 0, (rd.choice(data.unique_times) if with_pop else 0)
            else:
                idx = np.random.randint(len(clicked))           # :392-396
                p = clicked[idx]
                t = data.train_user_list_time[u][idx] if with_pop else 0
            while True:                                         # :397-401
                n = rd.choice(data.items)
                if n not in clicked:
                    break
            pos.append(p)
            neg.append(n)
            if with_pop:
                ppop.append(data.expo_popularity[p, t])         # :402-403
                npop.append(data.expo_popularity[n, t])
        yield (users, pos, neg, ppop, npop) if with_pop else (users, pos, neg)


def to_device_batch(batch, device):
    """Tuple of python lists / numpy arrays (sampler protocol) -> int32/float32 device tensors."""
    out = [torch.as_tensor(np.asarray(b, dtype=np.int32), device=device) for b in batch[:3]]
    if len(batch) == 5:
        out += [torch.as_tensor(np.asarray(b, dtype=np.float32), device=device) for b in batch[3:]]
    return tuple(out)


class DeviceSampler:
    """ahead: batches drawn per sampler launch (pda_sample_batches_dev; the reference's generator thread keeps a queue of
    batches as well, :178-220).  The batches are bit for bit those of ahead = 1 (one pda_sample_triplets launch per step); a
    batch handed out is a row view of the queue and stays valid until `ahead` further batches have been taken."""

    def __init__(self, data, device, with_pop: bool, seed: int = 2020, neg_range=None, ahead: int = 32):
        self.data, self.device, self.with_pop, self.seed = data, torch.device(device), with_pop, seed
        self.indptr, self.indices, self.slots = data.train_csr(self.device)
        pool = np.fromiter(data.train_user_list.keys(), dtype=np.int32)   # all_users = users with train rows
        self.pool = torch.from_numpy(pool).to(self.device)
        self.pop = None
        if with_pop:
            self.pop = torch.as_tensor(np.ascontiguousarray(data.expo_popularity, dtype=np.float32), device=self.device)
        self.neg_range = neg_range or (0, data.n_items)
        self.step = 0
        self.ahead = max(1, int(ahead))
        self._queue, self._left, self._calls = None, 0, 0
        # users are distinct inside a batch (pda_sample_triplets: a keyed permutation of the pool) as long as the batch is not
        # larger than the pool -- the contract the planned exact SGD step relies on (ops.bpr_step_plan)
        self.distinct_users = data.batch_size <= self.pool.numel()
        self.with_plan = False        # set by the trainer (--optimizer sgd): every batch comes with its pda_triplet_plan
        self.plan = None              # the plan of the batch handed out last

    def _refill(self):
        B, n = self.data.batch_size, self.ahead
        if self._queue is None or self._queue[0].shape != (n, B):
            mk = lambda dt: torch.empty((n, B), dtype=dt, device=self.device)
            # two queues in turn: a batch handed out stays valid while the next `ahead` are drawn and consumed
            self._queues = [(mk(torch.int32), mk(torch.int32), mk(torch.int32), mk(torch.float32) if self.with_pop else None,
                             mk(torch.float32) if self.with_pop else None) for _ in range(2)]
            self._ctr = torch.tensor([self.step, 0], dtype=torch.int64, device=self.device)      # (batch() has counted this one)
            self._calls = 0
            self._plans = None
        self._queue = self._queues[self._calls & 1]
        ops.sample_batches_into(self._queue, self.indptr, self.indices, seed=self.seed, step_dev=self._ctr, parity=self._calls & 1,
                                user_pool=self.pool, n_pool=self.pool.numel(), train_slots=self.slots if self.with_pop else None,
                                neg_range=self.neg_range, pop_matrix=self.pop)
        if self.with_plan:
            # the plans of the whole queue in ONE launch (one workgroup per batch), like the batches themselves
            if self._plans is None:
                nb = ops.triplet_plan_bytes(B)
                self._plans = [torch.empty((n, nb), dtype=torch.uint8, device=self.device) for _ in range(2)]
            ops.triplet_plan(self._queue[0], self._queue[1], self._queue[2], out=self._plans[self._calls & 1])
        self._calls += 1
        self._left = n

    def batch(self):
        self.step += 1
        if self.ahead == 1:
            u, p, n, pp, pn = ops.sample_triplets(self.indptr, self.indices, self.data.batch_size, seed=self.seed,
                                                  step=self.step, user_pool=self.pool, n_pool=self.pool.numel(),
                                                  train_slots=self.slots if self.with_pop else None,
                                                  neg_range=self.neg_range, pop_matrix=self.pop)
            self.plan = ops.triplet_plan(u, p, n)[0] if self.with_plan else None
            return (u, p, n, pp, pn) if self.with_pop else (u, p, n)
        if self._left == 0:
            self._refill()
        j = self.ahead - self._left
        self._left -= 1
        q = self._queue
        self.plan = self._plans[(self._calls - 1) & 1][j] if (self.with_plan and self._plans is not None) else None
        return (q[0][j], q[1][j], q[2][j], q[3][j], q[4][j]) if self.with_pop else (q[0][j], q[1][j], q[2][j])

    def __call__(self):
        """Zero-argument generator: one epoch of device-tensor batches."""
        for _ in range(n_batches(self.data)):
            yield self.batch()
