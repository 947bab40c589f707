python -m pytest tests/test_gpu_funnel.py -x -q 2>&1 | tail -4
python -m pytest tests/test_gpu_two_rank.py -x -q 2>&1 | tail -3
