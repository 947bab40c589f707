#!/bin/bash
# pipe-busy and clock of the huge geometry's loop variants (tools/ab_huge.sh): ab_huge_pmc.sh <tag> ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for t in "$@"; do
  if [ $t = base ]; then L=; else L=$R/pda_amd/csrc/ab/libpda_hip_$t.so; fi
  O=$R/gpurun_out/abpmc/$t; rm -rf $O; mkdir -p $O
  PDA_HIP_LIB=$L timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES --output-format csv -d $O -o p -- python tools/time_huge.py c3 262144 huge > $O/log.txt 2>&1
  python - $O $t <<'PY'
import csv, glob, sys, collections
O, t = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list); dur = []
for f in glob.glob(O + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sweep5" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(O + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sweep5" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
m = {k: sum(v) / len(v) for k, v in acc.items()}
d = sum(dur) / len(dur)
cyc = m["GRBM_GUI_ACTIVE"] / 8
print("%-8s sweep5 %.0f us  clock %.3f GHz  MFMA pipe busy %.1f %%  wait_inst_any/wave_cycles %.3f  wait_lds %.4f  valu insts %.3g  vmem cycles %.3g" % (
    t, d, cyc / d / 1e3, 100 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_LDS"] / m["SQ_WAVE_CYCLES"], m["SQ_INSTS_VALU"], m["SQ_INST_CYCLES_VMEM_RD"]))
PY
done
