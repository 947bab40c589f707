#!/bin/bash
# kernel-trace + PMC of the huge geometry at the headline block
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
PDA_SCORE_LISTS=huge rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python tools/time_huge.py c3 262144 huge > $O/kt.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $O/p1 -o p1 -- python tools/time_huge.py c3 262144 huge > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD --output-format csv -d $O/p2 -o p2 -- python tools/time_huge.py c3 262144 huge > $O/p2.log 2>&1
python - <<PY
import csv, glob, collections
for f in glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True):
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 8: print("stats", r["Name"][:80], r["Calls"], r["AverageNs"], r["Percentage"])
for tag in ("p1", "p2"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            if "sweep5" in k or "warm4" in k or "uprep5" in k:
                print(tag, k)
                for c, v in sorted(d.items()):
                    print("    %-28s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
    for f in glob.glob("$O/%s/**/*kernel_trace.csv" % tag, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in acc.items():
            if "sweep5" in k or "warm4" in k or "uprep5" in k:
                print(tag, "trace", k, "n=%d mean_us=%.1f" % (len(v), sum(v) / len(v)))
PY
