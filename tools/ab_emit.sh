#!/bin/bash
# timing / debugging variants of the emitting loop: tools/ab_emit.sh <tag> "<VAR=val ...>"  ->  pda_amd/csrc/ab/libpda_hip_<tag>.so (select with PDA_HIP_LIB)
set -e
cd "$(dirname "$0")/.."
mkdir -p pda_amd/csrc/ab
env $2 python tools/gen_v7_emit_loop_asm.py > pda_amd/csrc/ab/loop7_$1.h
cd pda_amd/csrc
sed "s#pda_v7_emit_loop_asm.h#loop7_$1.h#" pda_v7_funnel.h > ab/funnel_$1.h
sed "s#pda_v7_funnel.h#ab/funnel_$1.h#" pda_score_funnel.hip > ab_funnel_$1.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -ffp-contract=off $EXTRA -c ab_funnel_$1.hip -o ab/funnel_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libpda_hip_$1.so pda_score_topk.o pda_score_prep.o pda_score_topk_v3.o pda_score_topk_v4.o ab/funnel_$1.o pda_bpr_step.o pda_bpr_plan.o pda_bpr_plan_large.o pda_aux.o
rm -f ab/funnel_$1.o ab_funnel_$1.hip
echo built ab/libpda_hip_$1.so
