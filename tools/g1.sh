cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 300 2>&1 | tail -6
