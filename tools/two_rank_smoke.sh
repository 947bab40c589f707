#!/bin/bash
# Plumbing check of bench.py's N>1 path on a ONE-GPU box: two ranks share cuda:0 and talk over gloo (RCCL refuses two
# ranks on one device).  The numbers mean nothing; the point is that the sharded path runs and agrees with N=1.
export PDA_BENCH_ONE_GPU=1 MASTER_ADDR=127.0.0.1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 \
    bench.py --gpus 2 --steps 3 --warmup 1 "$@"
