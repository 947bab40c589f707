#!/bin/bash
# kernel trace of one funnel call on the small shapes (per-kernel average over the script's calls)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r6n}; mkdir -p $O; cd $R
for c in "c3 2048 20" "c2 50000 8"; do set -- $c
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/f_$1_$2 -o t -- python tools/time_funnel.py $1 $2 $3 > $O/f_$1_$2.log 2>&1
  echo "== $1 $2"; grep "raw head" $O/f_$1_$2.log | cut -c1-100
  python - <<PY
import csv, glob
for f in glob.glob("$O/f_$1_$2/**/*kernel_stats.csv", recursive=True):
    for i, r in enumerate(csv.DictReader(open(f))):
        if "anonymous" in r["Name"] and i < 24: print("%-86s calls %5s avg %8.1f us total %8.0f" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"])/1e3))
PY
done
