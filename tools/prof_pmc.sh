#!/bin/bash
# usage: tools_prof.sh <outdir-under-gpurun_out> ; runs 4 separate PMC passes of the eval bench (C3, 3 steps)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
CMD="python $R/bench.py --steps 3 --warmup 1 --no-train --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o p1 -- $CMD > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/p2 -o p2 -- $CMD > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p3 -o p3 -- $CMD > $O/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p4 -o p4 -- $CMD > $O/p4.log 2>&1
find $O -name "*.csv" | head -20
