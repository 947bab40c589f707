"""RCCL on real hardware: skipped on a one-GPU box, runs by itself the day two or more GPUs are visible.

Everything multi-rank in this repo has so far run over gloo (CPU tensors, or N ranks on ONE GPU: tests/test_dist_gloo.py,
tests/test_gpu_two_rank.py).  This test launches bench.py as two ranks on two GPUs with backend "nccl" (= RCCL on ROCm) -- the
all_to_all_single of the partial lists, the all_gather_into_tensor of the hot-item seeds, the all_reduce of the timing, all on
int64 / float32 device tensors -- and demands the lists of the one-rank run back.  Reference loop being sharded:
MF/train_new_api.py:780-794."""
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


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs two GPUs (the pod has one; the driver's 8-GPU node runs it)")
def test_two_ranks_over_rccl_return_the_one_rank_lists(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", PDA_BENCH_DUMP=str(tmp_path))
    env.pop("PDA_BENCH_ONE_GPU", None)
    common = ["--workload", "tiny", "--steps", "2", "--warmup", "1", "--eval-block", "2048"]
    one = _run([sys.executable, "bench.py", "--no-train", "--no-cpu-baseline"] + common, env)
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29531", "bench.py", "--gpus", "2"] + common, env)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["backend"] == "nccl" and two["config"]["ranks_seen"] == 2
    assert two["config"]["layout"]["item_shards"] == 2 and two["value"] > 0
    a = torch.load(os.path.join(tmp_path, "topk_w1_r0.pt"))
    b = torch.cat([torch.load(os.path.join(tmp_path, "topk_w2_r%d.pt" % r)) for r in range(2)])
    assert torch.equal(a, b)                                        # the early-terminating pass: seeded sweeps + the all-to-all
    da = torch.load(os.path.join(tmp_path, "topk_dense_w1_r0.pt"))
    db = torch.cat([torch.load(os.path.join(tmp_path, "topk_dense_w2_r%d.pt" % r)) for r in range(2)])
    assert torch.equal(da, db)                                      # the dense headline pass
