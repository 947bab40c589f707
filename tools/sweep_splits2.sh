#!/bin/bash
# the library's choice (ops.huge_splits) against the 256-user geometry and against neighbouring split counts
cd $GRAFT_REPO_ROOT
for u in 2048 4096 8192 24576 40960 49152 98304 163840 196608 229376; do python tools/time_huge.py c3 $u lds,auto 2>&1 | tail -2; done
python tools/time_huge.py c3 4096 huge 0 32 | tail -1
python tools/time_huge.py c3 98304 huge 0 5 | tail -1
python tools/time_huge.py c3 163840 huge 0 8 | tail -1
python tools/time_huge.py c3 196608 huge 0 1 | tail -1
python tools/time_huge.py c3 229376 huge 0 8 | tail -1
python tools/time_huge.py c3 65536 huge 0 8 | tail -1
for cfg in "c2 50000" "c1 47890"; do python tools/time_huge.py $cfg lds,auto | tail -2;  python tools/time_huge.py $cfg huge 0 8 | tail -1; python tools/time_huge.py $cfg huge 0 10 | tail -1; done
python tools/time_huge.py c3 262144 huge 0 1 3 | tail -1
python tools/time_huge.py c3 262144 huge 0 1 2 | tail -1
python tools/time_huge.py c3 262144 huge 0 1 4 | tail -1
python tools/shard_scaling.py
