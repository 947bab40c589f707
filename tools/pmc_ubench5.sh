#!/bin/bash
# PMC of the round-4 structural micro-benchmark (tools/ubench/mfma_struct5): matrix-pipe busy, clock, LDS activity per mapping
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/u1 -o u1 -- $R/tools/ubench/mfma_struct5 > $O/u1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA --output-format csv -d $O/u2 -o u2 -- $R/tools/ubench/mfma_struct5 > $O/u2.log 2>&1
python - <<PY
import csv, glob, collections
for tag in ("u1", "u2"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            print(tag, k)
            for c, v in sorted(d.items()):
                print("    %-28s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
    for f in glob.glob("$O/%s/**/*kernel_trace.csv" % tag, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in acc.items():
            print(tag, "trace", k, "n=%d mean_us=%.1f" % (len(v), sum(v) / len(v)))
PY
