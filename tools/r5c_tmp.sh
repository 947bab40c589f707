python tools/check_funnel.py small 2>&1 | grep -c "rows that differ 0"
python tools/check_funnel.py c3 262144 2>&1 | tail -5
python tools/time_funnel.py c3 262144 6 2>&1 | grep "funnel:" | sed 's/, kernel.*//'
