#!/bin/bash
# item splits behind the shared warm-up: where does the huge geometry with splits beat the 256-user geometry, and how many splits?
cd $GRAFT_REPO_ROOT
for cfg in "c2 50000" "c1 47890"; do
  for s in 1 2 3 4 5; do python tools/time_huge.py $cfg lds,huge 0 $s 2>&1 | tail -2; done
done
for u in 8192 16384 24576 32768 40960; do
  for s in 0 8 16 24 32; do python tools/time_huge.py c3 $u lds,huge 0 $s 2>&1 | tail -2; done
done
python tools/time_huge.py c3 65536 huge 0 3 2>&1 | tail -1
python tools/time_huge.py c3 98304 huge 0 2 2>&1 | tail -1
python tools/time_huge.py c3 163840 huge,wide 0 1 2>&1 | tail -2
