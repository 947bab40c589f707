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
    def __init__(self, data, device, with_pop: bool, seed: int = 2020, neg_range=None):
        self.data, self.device, self.with_pop, self.seed = data, torch.device(device), with_pop, seed
        self.indptr, self.indices, self.slots = data.train_csr(self.device)
        pool = np.fromiter(data.train_user_list.keys(), dtype=np.int32)   # all_users = users with train rows
        self.pool = torch.from_numpy(pool).to(self.device)
        self.pop = None
        if with_pop:
            self.pop = torch.as_tensor(np.ascontiguousarray(data.expo_popularity, dtype=np.float32), device=self.device)
        self.neg_range = neg_range or (0, data.n_items)
        self.step = 0

    def batch(self):
        self.step += 1
        u, p, n, pp, pn = ops.sample_triplets(self.indptr, self.indices, self.data.batch_size, seed=self.seed,
                                              step=self.step, user_pool=self.pool, n_pool=self.pool.numel(),
                                              train_slots=self.slots if self.with_pop else None,
                                              neg_range=self.neg_range, pop_matrix=self.pop)
        return (u, p, n, pp, pn) if self.with_pop else (u, p, n)

    def __call__(self):
        """Zero-argument generator: one epoch of device-tensor batches."""
        for _ in range(n_batches(self.data)):
            yield self.batch()
