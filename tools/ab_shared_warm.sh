#!/bin/bash
# shared warm-up (one exact warm-up per user) against one warm-up per item split (PDA_WARM_PER_SPLIT=1): the reference's 2 048-user blocks,
# mid-size blocks of the huge geometry, config 2
cd $GRAFT_REPO_ROOT
for v in "" 1; do
  export PDA_WARM_PER_SPLIT=$v
  echo "== PDA_WARM_PER_SPLIT='$v'"
  python tools/block2048.py c3 2>&1 | tail -2
  python tools/block2048.py c2 2>&1 | tail -2
  PDA_SCORE_LISTS= python tools/time_huge.py c3 65536 huge 0 4 2>&1 | tail -1
  python tools/time_huge.py c3 131072 huge 0 2 2>&1 | tail -1
  python tools/time_huge.py c3 32768 lds,huge 0 8 2>&1 | tail -2
  python tools/time_huge.py c2 50000 lds,huge 0 3 2>&1 | tail -2
done
