#!/usr/bin/env python
"""Copy the judged summaries of one tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/<name>_*."""
import csv, glob, io, os, subprocess, sys
tag, name = sys.argv[1], sys.argv[2]
src, dst = os.path.join("gpurun_out", tag), "profiles"
os.makedirs(dst, exist_ok=True)
rows = list(csv.reader(open(os.path.join(src, "stats", "bench_kernel_stats.csv"))))
with open(os.path.join(dst, name + "_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(rows[0])
    for r in rows[1:13]:
        r[0] = r[0][:160]
        w.writerow(r)
hl = os.path.join(src, "stats_headline", "bench_kernel_stats.csv")
if os.path.exists(hl):
    rows = list(csv.reader(open(hl)))
    with open(os.path.join(dst, name + "_kernel_stats_headline_only.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:7]:
            r[0] = r[0][:160]
            w.writerow(r)
pat = sys.argv[3] if len(sys.argv) > 3 else "score_topk"
pm = "# kernel pattern: %s\n" % pat + subprocess.check_output([sys.executable, "tools/pmc_summary.py", src, pat]).decode()
open(os.path.join(dst, name + "_pmc.txt"), "w").write(
    "# rocprofv3 --pmc passes (separate runs, tools/profile_round.sh) of: python bench.py [workload] --steps 3 --warmup 1 --no-train --no-cpu-baseline --headline-only\n"
    "# mean counter value per dispatch of the kernel named below, summed over the 8 XCDs; FETCH_SIZE/WRITE_SIZE in KiB\n"
    "# (gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x -- guides/MI355X_MICROARCH.md, HBM section)\n" + pm)
line = open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1]
ex = os.path.join(src, "bench_extras.json")
if os.path.exists(ex):
    open(os.path.join(dst, name + "_bench_extras.json"), "w").write(open(ex).read() + "\n")
open(os.path.join(dst, name + "_bench.json"), "w").write(line + "\n")
print(open(os.path.join(dst, name + "_pmc.txt")).read())
