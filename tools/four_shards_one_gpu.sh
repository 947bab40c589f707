#!/bin/bash
# bench.py as FOUR item shards of config 3 on ONE GPU over gloo (PDA_BENCH_ONE_GPU=1), item shards only: the replicated-hot-items path of
# pda_amd.dist at full scale with the real kernels; the ranks' lists against the one-rank run (timings over gloo mean nothing)
cd $GRAFT_REPO_ROOT
D=/tmp/dump4; rm -rf $D; mkdir -p $D
export PDA_BENCH_ONE_GPU=1 MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0 PDA_BENCH_DUMP=$D
python bench.py --steps 2 --warmup 1 --no-train --no-cpu-baseline --no-per-config --headline-only --eval-block 131072 > $D/one.json 2> $D/one.err || tail -5 $D/one.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 4 --user-groups 1 --steps 2 --warmup 1 \
  --no-train --no-cpu-baseline --no-per-config --headline-only --eval-block 131072 > $D/four.json 2> $D/four.err || tail -20 $D/four.err
python - <<PY
import json, torch, glob
one = json.loads([l for l in open("$D/one.json") if l.startswith("{")][-1]); four = json.loads([l for l in open("$D/four.json") if l.startswith("{")][-1])
print("one rank:", one["ms_per_step"], "ms per step; four ranks on one GPU over gloo:", four["ms_per_step"], four["config"]["layout"], four["config"]["item_shard_path"][:60])
a = torch.load("$D/topk_dense_w1_r0.pt"); b = torch.cat([torch.load("$D/topk_dense_w4_r%d.pt" % r) for r in range(4)])
print("dense lists of the last step: shapes", tuple(a.shape), tuple(b.shape), "equal:", bool(torch.equal(a, b)))
PY
