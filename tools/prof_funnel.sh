#!/bin/bash
# rocprofv3 kernel stats of the funnel on a bench workload: tools/prof_funnel.sh [workload=c3] [users=262144] [tag]
cd "$(dirname "$0")/.."
WL=${1:-c3}; BU=${2:-262144}; TAG=${3:-funnel}
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o run -- python tools/check_funnel.py $WL $BU > gpurun_out/prof_$TAG/log.txt 2>&1
tail -4 gpurun_out/prof_$TAG/log.txt
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-90s calls %5s avg %10.1f us total %8.2f ms  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
