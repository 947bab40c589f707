python -m pytest tests/test_gpu_funnel.py -x -q 2>&1 | tail -2
python tools/check_funnel.py small 2>&1 | grep -c "rows that differ 0"
python tools/check_funnel.py c3 262144 2>&1 | tail -3
bash tools/prof_funnel.sh c3 262144 r5t 2>&1 | grep "7_kernel"
python tools/check_funnel.py c3 20000 2>&1 | tail -3
