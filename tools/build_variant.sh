#!/bin/bash
# A/B builds of the v4 kernel file: tools/build_variant.sh <tag> "<extra hipcc flags>"  ->  pda_amd/csrc/ab/libpda_hip_<tag>.so
# (select it with PDA_HIP_LIB=<path>; the other objects are the ones of the regular build)
set -e
cd "$(dirname "$0")/../pda_amd/csrc"
mkdir -p ab
/opt/rocm/bin/hipcc $2 --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -ffp-contract=off -c ${SRC:-pda_score_topk_v4.hip} -o ab/v4_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libpda_hip_$1.so pda_score_topk.o pda_score_prep.o pda_score_topk_v3.o ab/v4_$1.o pda_score_funnel.o pda_bpr_step.o pda_bpr_plan.o pda_bpr_plan_large.o pda_aux.o
rm ab/v4_$1.o
echo built ab/libpda_hip_$1.so
