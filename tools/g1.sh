cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_score_topk.py -x -q -m gpu -k "c1_shape" --timeout 1000 2>&1 | tail -5
