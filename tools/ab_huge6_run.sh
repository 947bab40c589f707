#!/bin/bash
# runs the timing-only variants built by tools/ab_huge6.sh (c3, 262 144 users): tools/ab_huge6_run.sh <tag> ...
for t in base "$@" base; do
  if [ $t = base ]; then L=; else L=pda_amd/csrc/ab/libpda_hip_$t.so; fi
  printf "%-10s " $t; PDA_HIP_LIB=$L timeout 120 python tools/time_huge.py c3 262144 huge 2>&1 | tail -n 1
done
