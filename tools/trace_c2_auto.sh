cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6d; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2pop -o t -- python tools/time_huge.py c2 50000 auto > $O/c2pop.log 2>&1
grep -v "^[EW]2026" $O/c2pop.log | tail -2
python - <<PY
import csv, glob
for f in glob.glob("$O/c2pop/**/*kernel_stats.csv", recursive=True):
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 12: print("%-100s calls %5s avg %9.1f us  %5s %%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
