#!/usr/bin/env python
"""Summarise rocprofv3 counter_collection CSVs: mean counter value per dispatch for kernels matching a pattern."""
import csv, glob, sys, collections
root, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "score_topk")
for f in sorted(glob.glob(root + "/**/*_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("%-34s n=%3d mean=%.6g" % (k, len(v), sum(v) / len(v)))
for f in sorted(glob.glob(root + "/**/*_kernel_trace.csv", recursive=True))[:1]:
    d = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
    print("kernel_trace: n=%d mean_us=%.1f" % (len(d), sum(d) / len(d) / 1e3))
