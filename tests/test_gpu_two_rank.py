"""The N > 1 path of bench.py with REAL kernels: two ranks under torch.distributed.run on the one GPU of the test box
(PDA_BENCH_ONE_GPU=1: both ranks on cuda:0, gloo instead of RCCL -- RCCL refuses two ranks on one device).  Checks that
the item-sharded evaluation starts, exchanges the partial lists (all-to-all orchestration of pda_amd.dist), prints the
contract line -- and that the sharded result equals the single-rank result on the same users.  RCCL itself stays
unexecuted until the driver's multi-GPU run."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env):
    """Runs bench.py; checks what the driver ingests (the last stdout line: compact JSON) and returns the FULL record (bench_extras)."""
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        extras = os.path.join(td, "extras.json")
        p = subprocess.run(cmd + ["--extras-path", extras], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
        assert len([ln for ln in lines if ln.startswith("{")]) == 1, p.stdout[-2000:]
        line = json.loads(lines[-1])
        assert len(lines[-1]) < 8192
        full = json.load(open(extras))
    assert line["value"] == full["value"] and line["n_gpus"] == full["n_gpus"] and line["config"]["layout"] == full["config"]["layout"]
    return full


def test_bench_two_ranks_on_one_gpu_matches_one_rank(tmp_path):
    # (PDA_HOT_ITEMS_MIN_SHARDS=2: the replicated-hot-items path of pda_amd.dist, by default taken from three item shards on)
    env = dict(os.environ, PDA_BENCH_ONE_GPU="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PDA_BENCH_DUMP=str(tmp_path), PDA_HOT_ITEMS_MIN_SHARDS="2")
    one = _run([sys.executable, "bench.py", "--workload", "tiny", "--steps", "2", "--warmup", "1", "--no-train", "--no-cpu-baseline",
                "--eval-block", "2048"], env)
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29517", "bench.py", "--gpus", "2", "--workload", "tiny", "--steps", "2", "--warmup", "1",
                "--eval-block", "2048"], env)
    for line, n in ((one, 1), (two, 2)):
        assert line["n_gpus"] == n and line["unit"] == "users/s" and line["value"] > 0 and line["scaling"] == "strong"
        assert line["roofline"]["bound"] == "mfma" and line["steps"] == 2
    # the lists of the last step: rank r of the two-rank run holds the rows of ITS slice of the users
    a = torch.load(os.path.join(tmp_path, "topk_w1_r0.pt"))
    b = torch.cat([torch.load(os.path.join(tmp_path, "topk_w2_r%d.pt" % r)) for r in range(2)])
    assert a.shape == b.shape == (2048, 50)
    assert torch.equal(a, b)
    # ... and of the DENSE headline pass, which for N > 1 goes through the replicated-hot-items path (pda_amd.dist._topk_blocks_hot)
    da = torch.load(os.path.join(tmp_path, "topk_dense_w1_r0.pt"))
    db = torch.cat([torch.load(os.path.join(tmp_path, "topk_dense_w2_r%d.pt" % r)) for r in range(2)])
    assert torch.equal(da, db) and torch.equal(da, a)
    # four ranks = four item shards (the default since round 5: BASELINE config 4's layout; replicated hot items from three shards on): rank order = user order
    four = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                 "--master-port", "29519", "bench.py", "--gpus", "4", "--workload", "tiny", "--steps", "2", "--warmup", "1",
                 "--eval-block", "2048"], env)
    assert four["n_gpus"] == 4 and four["config"]["layout"] == {"user_groups": 1, "item_shards": 4, "users_per_rank_and_step": 2048,
                                                                 "items_per_rank": 768}
    grid = four["user_groups_grid"]                                  # ... and beside it the two-dimensional layout: 2 user groups x 2 item shards
    assert grid["value"] > 0 and grid["layout"] == {"user_groups": 2, "item_shards": 2, "users_per_rank_and_step": 1024, "items_per_rank": 1504}
    c = torch.cat([torch.load(os.path.join(tmp_path, "topk_w4_r%d.pt" % r)) for r in range(4)])
    assert torch.equal(a, c)
    dc = torch.cat([torch.load(os.path.join(tmp_path, "topk_dense_w4_r%d.pt" % r)) for r in range(4)])
    assert torch.equal(da, dc)


def test_bench_eight_ranks_default_layout_on_one_gpu(tmp_path):
    """What the driver's scaling run launches at N = 8, on one GPU over gloo: the default layout -- eight item shards (BASELINE
    config 4), the replicated hot items, the all-to-all of the partial lists -- and the union of the ranks' lists equals the
    single-rank lists; the two-dimensional layout (4 user groups x 2 item shards) is timed beside it."""
    env = dict(os.environ, PDA_BENCH_ONE_GPU="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PDA_BENCH_DUMP=str(tmp_path))
    one = _run([sys.executable, "bench.py", "--workload", "tiny", "--steps", "2", "--warmup", "1", "--no-train", "--no-cpu-baseline",
                "--eval-block", "2048"], env)
    eight = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                  "--master-port", "29523", "bench.py", "--gpus", "8", "--workload", "tiny", "--steps", "2", "--warmup", "1",
                  "--eval-block", "2048"], env)
    assert eight["n_gpus"] == 8 and eight["value"] > 0 and eight["scaling"] == "strong"
    lay = eight["config"]["layout"]
    assert lay["user_groups"] == 1 and lay["item_shards"] == 8 and lay["users_per_rank_and_step"] == 2048
    assert eight["ordered_sweep"]["value"] > 0                      # the early-terminating pass ran on every rank
    grid = eight["user_groups_grid"]
    assert grid["value"] > 0 and grid["layout"]["user_groups"] == 4 and grid["layout"]["item_shards"] == 2 and grid["early_terminating_sweep"]["value"] > 0
    assert "replicated hot rows" in eight["config"]["item_shard_path"]
    a = torch.load(os.path.join(tmp_path, "topk_w1_r0.pt"))
    c = torch.cat([torch.load(os.path.join(tmp_path, "topk_w8_r%d.pt" % r)) for r in range(8)])
    assert a.shape == c.shape and torch.equal(a, c) and one["n_gpus"] == 1
    dc = torch.cat([torch.load(os.path.join(tmp_path, "topk_dense_w8_r%d.pt" % r)) for r in range(8)])
    assert torch.equal(torch.load(os.path.join(tmp_path, "topk_dense_w1_r0.pt")), dc)


def test_config4_eight_item_shards_at_full_size_on_one_gpu(tmp_path):
    """BASELINE config 4 (1M users x 200k items, d = 128, the catalogue item-sharded over EIGHT ranks: 25 000 items per rank, 262 144 users per
    step, replicated hot items, the all-to-all of the partial lists) with the real kernels at full size -- eight processes on the ONE GPU of the
    test box over gloo (RCCL needs eight devices: tests/test_gpu_rccl.py, the driver's scaling run).  The union of the ranks' lists of a step
    equals the one-rank lists row for row, for the dense headline pass and for the early-terminating pass."""
    env = dict(os.environ, PDA_BENCH_ONE_GPU="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", PDA_BENCH_DUMP=str(tmp_path),
               PDA_BENCH_WATCHDOG="800")         # (a hang ends in every thread's stack on stderr instead of a silent timeout)
    common = ["--workload", "c3", "--steps", "2", "--warmup", "1", "--no-train", "--no-cpu-baseline", "--no-per-config", "--eval-block", "262144"]
    one = _run([sys.executable, "bench.py"] + common, env)
    eight = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                  "--master-port", "29541", "bench.py", "--gpus", "8"] + common, env)
    lay = eight["config"]["layout"]
    assert lay["user_groups"] == 1 and lay["item_shards"] == 8 and lay["users_per_rank_and_step"] == 262144, lay
    assert 25000 <= lay["items_per_rank"] <= 25088, lay          # (rank 0's shard: 200 000 / 8, cut at a multiple of 64 items)
    assert "replicated hot rows" in eight["config"]["item_shard_path"] and one["config"]["layout"]["item_shards"] == 1
    for name in ("topk_dense", "topk"):
        a = torch.load(os.path.join(tmp_path, "%s_w1_r0.pt" % name))
        b = torch.cat([torch.load(os.path.join(tmp_path, "%s_w8_r%d.pt" % (name, r))) for r in range(8)])
        assert a.shape == b.shape == (262144, 50), (name, a.shape, b.shape)
        assert torch.equal(a, b), name
