#!/bin/bash
# runs the timing-only variants built by tools/ab_huge6.sh / tools/ab_huge.sh (c3, 262 144 users)
for t in base v6notest v6bare v6pure; do
  if [ $t = base ]; then L=; else L=pda_amd/csrc/ab/libpda_hip_$t.so; fi
  echo "== $t"; PDA_HIP_LIB=$L timeout 120 python tools/time_huge.py c3 262144 huge 2>&1 | tail -n 1
done
for t in base v5notest v5bare; do
  if [ $t = base ]; then L=; else L=pda_amd/csrc/ab/libpda_hip_$t.so; fi
  echo "== $t (32x32x16)"; PDA_HIP_LIB=$L timeout 120 python tools/time_huge.py c3 262144 huge32 2>&1 | tail -n 1
done
