#!/bin/bash
# needs a -DPDA_ABLATION build.  bits: 1 drop candidates, 2 no test, 4 no history, 8 no global tile loads / LDS stores,
# 16 no LDS reads (B fragments from registers), 32 no barriers
for v in ${@:-0 7 15 23 31 39 47 63}; do
  echo -n "ABL=$v: "; PDA_SCORE_PRUNE=0 PDA_ABLATE=$v python bench.py --no-train --no-cpu-baseline --steps 6 --warmup 2 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"
done
