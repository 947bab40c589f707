#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --steps 5 --warmup 1 --no-train --no-cpu-baseline > $O/log.txt 2>&1
head -6 $O/stats/b_kernel_stats.csv | cut -c1-150
