python tools/time_emit.py time c3 262144 2>&1 | grep "users,"
