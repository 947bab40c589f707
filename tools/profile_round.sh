#!/bin/bash
# usage (on the GPU box via gpurun): tools/profile_round.sh <tag> [bench.py arguments, e.g. --workload c5shard --no-per-config]
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command  -> gpurun_out/<tag>/stats
#   2. four PMC passes of the eval-only bench (counters in separate runs, no other trace domain)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
shift; X="$*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py $X > $O/bench_under_rocprof.log 2>&1
python $R/bench.py $X --extras-path $O/bench_extras.json > $O/bench.json 2> $O/bench.err
python $R/tools/check_contract_line.py $O/bench.json | tee $O/contract_line_check.txt
# the early-terminating sweep launches the SAME kernel template as the headline sweep: a second summary with the headline alone
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_headline -o bench -- python $R/bench.py $X --headline-only --no-train --no-cpu-baseline --extras-path /dev/null > $O/bench_headline_under_rocprof.log 2>&1
CMD="python $R/bench.py $X --steps 3 --warmup 1 --no-train --no-cpu-baseline --headline-only --extras-path /dev/null"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o p1 -- $CMD > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/p2 -o p2 -- $CMD > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p3 -o p3 -- $CMD > $O/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p4 -o p4 -- $CMD > $O/p4.log 2>&1
find $O -name "*.csv" | head -30
