#!/bin/bash
# usage (GPU box): tools/pmc_quick.sh <tag>   -- two PMC passes over the headline-only bench (instruction mix of the sweep kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
CMD="python $R/bench.py --steps 3 --warmup 1 --no-train --no-cpu-baseline --headline-only"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o p1 -- $CMD > $O/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/p2 -o p2 -- $CMD > $O/p2.log 2>&1
python $R/tools/pmc_summary.py $O "score_topk_v3_kernel<128, 1, true"
