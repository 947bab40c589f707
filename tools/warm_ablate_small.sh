#!/bin/bash
# round 6: where a warm-up workgroup's time goes in the single-round cases (C2: 391 workgroups of 128 users; 2 048-user blocks: 16) -- the -DPDA_W4_ABL
# builds of tools/build_variant.sh (1 no history walk, 2 no selection / emission, 4 no final sort, 8 no MFMA; results are wrong by construction)
cd $GRAFT_REPO_ROOT
for v in base w4abl1 w4abl2 w4abl4 w4abl6 w4abl7 w4abl8 w4abl15; do
  if [ $v = base ]; then unset PDA_HIP_LIB; else export PDA_HIP_LIB=$PWD/pda_amd/csrc/ab/libpda_hip_$v.so; fi
  [ $v = base ] || [ -f "$PDA_HIP_LIB" ] || continue
  for c in "c2 50000" "c3 2048" "c3 262144"; do set -- $c; echo "$v: $(timeout 120 python tools/time_warm.py $1 $2 2>&1 | grep warm-up)"; done
done
