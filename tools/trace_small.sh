#!/bin/bash
# kernel traces of the shapes whose time is fixed cost: the reference's 2 048-user block protocol, the C2 / C1 dense sweep, one rank's item shard
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b2048 -o t -- python tools/block2048.py c3 > $O/b2048.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -o t -- python tools/time_huge.py c2 50000 lds > $O/c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/shard -o t -- python tools/time_huge.py c3 262144 huge 25000 > $O/shard.log 2>&1
for n in b2048 c2 shard; do
  echo "== $n"; cat $O/$n.log | tail -4
  python - <<PY
import csv, glob
for f in glob.glob("$O/$n/**/*kernel_stats.csv", recursive=True):
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 14: print("%-90s calls %5s avg %9.1f us  %5s %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done
