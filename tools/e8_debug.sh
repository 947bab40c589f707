cd $GRAFT_REPO_ROOT
( time timeout 1000 python -m pytest tests/test_gpu_two_rank.py -x -q -m gpu -k "config4_eight" 2>&1 | tail -15 ) 2>&1 | tail -20
( time timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "c5_shard" 2>&1 | tail -15 ) 2>&1 | tail -20
timeout 600 python -m pytest tests/test_gpu_end_to_end.py -x -q -m gpu -k "testing_and_predict" 2>&1 | tail -8
