cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bpr_step.py -x -q -m gpu -k "exact_sgd or sgd_fused or end_to" 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_end_to_end.py tests/test_abi.py -x -q 2>&1 | tail -3
