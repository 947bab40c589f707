#!/bin/bash
for v in 0 1; do
  echo -n "ABL=$v: "; PDA_ABLATE=$v python bench.py --no-train --no-cpu-baseline --steps 6 --warmup 2 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['roofline']['frac'])"
done
