#!/bin/bash
# BASELINE config 4 on ONE GPU: bench.py as EIGHT item shards of config 3 (25 000 items per rank, 262 144 users per step, replicated hot items)
# over gloo (PDA_BENCH_ONE_GPU=1); the ranks' lists against the one-rank run (timings of eight processes sharing one GPU mean nothing).
# The same check runs in the GPU suite: tests/test_gpu_two_rank.py::test_config4_eight_item_shards_at_full_size_on_one_gpu
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_two_rank.py -x -q -m gpu -k "config4_eight" 2>&1 | tail -5
