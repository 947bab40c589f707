#!/bin/bash
# timing-only A/B builds of the huge geometry's loop: tools/ab_huge.sh <tag> <V5_VARIANT list>  ->  pda_amd/csrc/ab/libpda_hip_<tag>.so
# (results of these builds are WRONG by construction -- they answer "what does this part of the loop cost")
set -e
cd "$(dirname "$0")/.."
mkdir -p pda_amd/csrc/ab
V5_VARIANT=$2 python tools/gen_v5_loop_asm.py > pda_amd/csrc/ab/loop_$1.h
tools/build_variant.sh $1 "-DPDA_V5_LOOP_HEADER=\"ab/loop_$1.h\""
