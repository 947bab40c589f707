#!/bin/bash
# usage (GPU box): tools/pmc_v4.sh <tag> <variant tags...>   -- PMC passes over tools/time_variants.py (one variant per run), sweep4 kernel only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift
for v in "$@"; do
  O=$R/gpurun_out/$tag/$v; mkdir -p $O
  CMD="python $R/tools/time_variants.py $v order 1"
  ( cd $R && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o p1 -- $CMD > $O/p1.log 2>&1 )
  ( cd $R && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/p2 -o p2 -- $CMD > $O/p2.log 2>&1 )
  echo "== $v"; python $R/tools/pmc_summary.py $O "sweep4_kernel<128"
done
