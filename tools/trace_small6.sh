#!/bin/bash
# round 6: kernel traces of the shapes that are all fixed cost -- C2 / C1 both heads, the reference's 2 048-user blocks (both heads)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2pop -o t -- python tools/time_huge.py c2 50000 huge > $O/c2pop.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2raw -o t -- python tools/time_funnel.py c2 50000 6 > $O/c2raw.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c1raw -o t -- python tools/time_funnel.py c1 47890 6 > $O/c1raw.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b2048raw -o t -- python tools/time_funnel.py c3 2048 20 > $O/b2048raw.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b2048pop -o t -- python tools/block2048.py c3 > $O/b2048pop.log 2>&1
for n in c2pop c2raw c1raw b2048raw b2048pop; do
  echo "== $n"; tail -3 $O/$n.log
  python - <<PY
import csv, glob
for f in glob.glob("$O/$n/**/*kernel_stats.csv", recursive=True):
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 16: print("%-100s calls %5s avg %9.1f us  %5s %%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done
