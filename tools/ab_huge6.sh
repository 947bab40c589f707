#!/bin/bash
# timing-only A/B build of the huge geometry's 16 x 16 x 32 loop: tools/ab_huge6.sh <tag> <V5_VARIANT list>  ->  pda_amd/csrc/ab/libpda_hip_<tag>.so
# (results of these builds are WRONG by construction -- they answer "what does this part of the loop cost")
set -e
cd "$(dirname "$0")/.."
mkdir -p pda_amd/csrc/ab
V5_VARIANT=$2 python tools/gen_v6_loop_asm.py > pda_amd/csrc/ab/loop6_$1.h
tools/build_variant.sh $1 "-DPDA_V6_LOOP_HEADER=\"ab/loop6_$1.h\""
