cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_score_topk.py -x -q -m gpu -k "seeded" --timeout 600 2>&1 | tail -40
