cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bpr_step.py tests/test_gpu_bench_contract.py -x -q -m gpu --timeout 600 2>&1 | tail -3
