"""CPU suite, part 4: the N>1 evaluation path under gloo, world_size 2.  The HIP entry points cannot run here, so
the orchestration (shard ranges, local merge, ONE all-gather, final merge, set_popularity) is exercised with
test doubles that answer through the CPU oracle; the result must equal the unsharded oracle bit for bit."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import c_oracle
from oracle import pda_oracle as po


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pack(vals, idxs):
    hi = vals.view(np.uint32).astype(np.uint64)
    ordb = np.where(hi & np.uint64(0x80000000), (~hi) & np.uint64(0xFFFFFFFF), hi | np.uint64(0x80000000))
    return ((ordb << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - idxs.astype(np.uint64))).view(np.int64)


def _unpack(keys):
    k = keys.view(np.uint64)
    hi = (k >> np.uint64(32)).astype(np.uint32)
    bits = np.where(hi & np.uint32(0x80000000), hi ^ np.uint32(0x80000000), ~hi).astype(np.uint32)
    return bits.view(np.float32), (np.uint32(0xFFFFFFFF) - (k & np.uint64(0xFFFFFFFF)).astype(np.uint32)).astype(np.int32)


def score_double(U, I_shard, users, K, head, pop_shard, hist, item_offset, n_splits):
    """Stand-in for ops.score_topk_keys: same contract ([n_splits, Bu, K] packed keys), computed by the oracle."""
    bip = bix = None
    if hist is not None:                      # hist = CSR by user id (replicated); the oracle wants block rows
        ip, ix = hist
        u = users.numpy()
        bip = np.zeros(len(u) + 1, np.int64)
        bip[1:] = np.cumsum(ip[u + 1] - ip[u])
        bix = np.concatenate([ix[ip[x]:ip[x + 1]] for x in u]).astype(np.int32)
    pop_full = None
    if pop_shard is not None:
        pop_full = np.zeros(item_offset + I_shard.shape[0], np.float32)
        pop_full[item_offset:] = pop_shard.numpy()
    I_full = np.zeros((item_offset + I_shard.shape[0], I_shard.shape[1]), np.float32)
    I_full[item_offset:] = I_shard.numpy()
    idx, val = c_oracle.score_topk(U.numpy(), I_full, users.numpy(), K, head, pop_full, bip, bix, item_offset=item_offset,
                                   n_items_local=I_shard.shape[0], order=1)
    return torch.from_numpy(_pack(val + np.float32(0), idx))[None]


def merge_double(keys, users, hist, want="idx_val"):
    v, i = _unpack(keys.numpy())
    mi, mv = po.merge_partial_topk(v, i, keys.shape[2])
    if want == "keys":
        return torch.from_numpy(_pack(mv, mi))
    return torch.from_numpy(mi), torch.from_numpy(mv)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pda_amd.dist import ItemShardedTopK
    rng = np.random.default_rng(4)                                  # same data on every rank (replicated inputs)
    nU, nI, d, K = 60, 777, 32, 50
    U = (rng.standard_normal((nU, d)) * 0.1).astype(np.float32)
    I = (rng.standard_normal((nI, d)) * 0.1).astype(np.float32)
    pop = (rng.uniform(0, 1, nI) ** 0.22).astype(np.float32)
    rows = [np.sort(rng.integers(0, nI, rng.integers(0, 25))).astype(np.int32) for _ in range(nU)]
    ip = np.zeros(nU + 1, np.int64)
    ip[1:] = np.cumsum([len(r) for r in rows])
    ix = np.concatenate(rows)
    ev = ItemShardedTopK.from_full_tables(torch.from_numpy(U), torch.from_numpy(I), torch.from_numpy(pop), rank, world,
                                          score_fn=score_double, merge_fn=merge_double)
    assert ev.I_shard.shape[0] < nI and ev.item_offset == (0 if rank == 0 else ev.item_offset)
    blocks = [torch.arange(0, 32, dtype=torch.int32), torch.arange(32, 60, dtype=torch.int32)]
    out = []
    for users, (idx, val) in zip(blocks, ev.topk_blocks(blocks, K, 1, (ip, ix))):
        u = users.numpy()
        bip = np.zeros(len(u) + 1, np.int64)
        bip[1:] = np.cumsum([len(rows[x]) for x in u])
        bix = np.concatenate([rows[x] for x in u])
        ridx, rval = c_oracle.score_topk(U, I, u, K, 1, pop, bip, bix, order=1)
        out.append(bool(np.array_equal(idx.numpy(), ridx) and np.array_equal(val.numpy(), rval)))
    # a new popularity vector is re-sliced per shard (evaluation.set_testing_popularity)
    pop2 = (pop * 0.5).astype(np.float32)
    ev.set_popularity(torch.from_numpy(pop2))
    idx, val = ev.topk(blocks[0], K, 1, None)
    ridx, rval = c_oracle.score_topk(U, I, blocks[0].numpy(), K, 1, pop2, order=1)
    out.append(bool(np.array_equal(idx.numpy(), ridx)))
    q.put((rank, out, ev.item_offset, ev.I_shard.shape[0]))
    dist.barrier()
    dist.destroy_process_group()


def test_item_sharded_eval_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    assert all(all(r[1]) for r in res), res
    (o0, n0), (o1, n1) = (res[0][2], res[0][3]), (res[1][2], res[1][3])
    assert o0 == 0 and o1 == n0 and n0 + n1 == 777 and n0 % 32 == 0
