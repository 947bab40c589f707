#!/bin/bash
# PMC passes of the train-step kernels (config 2, B = 2 048; tools/time_train.py replays HIP graphs of 64 steps): usage via gpurun: bash tools/pmc_train.sh <tag>
# separate --pmc passes, kernel-trace only beside them; summary: python tools/pmc_summary.py gpurun_out/<tag> <kernel substring>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
CMD="python $R/tools/time_train.py c2 2048"
cd $R
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/p1 -o p1 -- $CMD > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR --output-format csv -d $O/p2 -o p2 -- $CMD > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p3 -o p3 -- $CMD > $O/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p4 -o p4 -- $CMD > $O/p4.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/p5 -o p5 -- $CMD > $O/p5.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/p6 -o p6 -- $CMD > $O/p6.log 2>&1
tail -n 12 $O/p1.log
for k in "bpr_step_kernel<64" "plan_triplets_kernel<64" "plan_items_kernel<64" "sgd_apply"; do echo "== $k"; python tools/pmc_summary.py $O "$k"; done
