#!/bin/bash
# PMC passes of the raw head's sweep (C3, 262 144 users, visiting order by norm: the many-candidates geometry); usage via gpurun: bash tools/pmc_raw_head.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
CMD="python $R/tools/time_v4.py c3 262144 0 v4"
cd $R
ONLY_ORDER=1 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o p1 -- $CMD > $O/p1.log 2>&1
ONLY_ORDER=1 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/p2 -o p2 -- $CMD > $O/p2.log 2>&1
ONLY_ORDER=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p3 -o p3 -- $CMD > $O/p3.log 2>&1
ONLY_ORDER=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p4 -o p4 -- $CMD > $O/p4.log 2>&1
grep head $O/p1.log
