#!/bin/bash
# round 6: the funnel's schedule on the small catalogues (C1 / C2: all fixed cost per launch) and the reference's 2 048-user blocks of C3
cd $GRAFT_REPO_ROOT
for wl in "c2 50000" "c1 47890" "c3 2048"; do
  set -- $wl
  for t in "default" "1e-6,8,64,4|2,6,20" "1e-6,8,64,4|2,6,40" "1e-6,16,64,4|2,6,80" "1e-6,8,64,4|4,6,40" "1e-6,4,64,4|4,6,20" "1e-6,4,64,4|2,3,40" "1e-6,8,96,4|2,4,60"; do
    if [ "$t" = "default" ]; then unset FUNNEL_TUNE FUNNEL_TUNE2; else export FUNNEL_TUNE="${t%%|*}" FUNNEL_TUNE2="${t##*|}"; fi
    echo "== $1 $2 tune=$t"
    timeout 120 python tools/time_funnel.py $1 $2 8 2>&1 | grep -E "schedule|raw head" | cut -c1-230
  done
done
