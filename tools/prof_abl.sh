#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
for v in 0 1 3; do
PDA_ABLATE=$v rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $O/a$v -o a$v -- python $R/bench.py --steps 3 --warmup 1 --no-train --no-cpu-baseline > $O/a$v.log 2>&1
PDA_ABLATE=$v rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/b$v -o b$v -- python $R/bench.py --steps 3 --warmup 1 --no-train --no-cpu-baseline > $O/b$v.log 2>&1
done
