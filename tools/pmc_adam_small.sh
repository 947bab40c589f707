#!/bin/bash
# round 6: HBM traffic of the reference-faithful Adam step on the C2 tables (pda_adam_step_f32: bpr_step_kernel + adam_dense_sweep4_kernel), resident
# against streaming cache policy: FETCH_SIZE / WRITE_SIZE in separate passes, kernel-trace only beside them.  usage via gpurun: bash tools/pmc_adam_small.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
CMD="python $R/tools/time_adam_small.py c2 512"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o t -- $CMD > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p3 -o p3 -- $CMD > $O/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p4 -o p4 -- $CMD > $O/p4.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/p5 -o p5 -- $CMD > $O/p5.log 2>&1
grep -v "^[EW]2026" $O/stats.log | tail -12
for k in "adam_dense_sweep4_kernel<false" "adam_dense_sweep4_kernel<true" "adam_dense_sweep3_kernel" "bpr_step_kernel<64"; do echo "== $k"; python tools/pmc_summary.py $O "$k"; done
