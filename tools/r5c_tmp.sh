python tools/check_funnel.py small 2>&1 | grep -c "rows that differ 0"
python tools/check_funnel.py c3 262144 2>&1 | tail -4
bash tools/prof_funnel.sh c3 262144 r5l 2>&1 | grep "7_kernel"
python tools/check_funnel.py c2 2>&1 | tail -3
