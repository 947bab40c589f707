cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
FUNNEL_TUNE=1e-3,4,64,4 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r6b -o run -- python tools/time_funnel.py c5shard 262144 2 > gpurun_out/prof_r6b.log 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_r6b/run_kernel_stats.csv")))
for r in rows[:12]:
    print("%-100s calls %5s avg %10.1f us total %8.2f ms" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
