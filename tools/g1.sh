cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_score_topk.py -x -q -m gpu -k "k4" --timeout 120 2>&1 | tail -4
timeout 900 python tools/time_v4.py c3 65536 10 v3,v4 2>&1 | tail -12
