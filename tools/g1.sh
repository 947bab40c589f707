cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
