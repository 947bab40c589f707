#!/usr/bin/env python
"""Summarise rocprofv3 counter_collection CSVs: mean counter value per dispatch for kernels matching a pattern."""
import csv, glob, sys, collections
root, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "score_topk")
tot = {}
for f in sorted(glob.glob(root + "/**/*_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("%-34s n=%3d mean=%.6g" % (k, len(v), sum(v) / len(v)))
        tot[k] = sum(v) / len(v)
for f in sorted(glob.glob(root + "/**/*_kernel_trace.csv", recursive=True))[:1]:
    d = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
    print("kernel_trace: n=%d mean_us=%.1f" % (len(d), sum(d) / len(d) / 1e3))
    us = sum(d) / len(d) / 1e3
    if "GRBM_GUI_ACTIVE" in tot and "SQ_VALU_MFMA_BUSY_CYCLES" in tot:
        cyc = tot["GRBM_GUI_ACTIVE"] / 8.0                      # shader cycles of one launch (the counter is summed over the 8 XCDs)
        print("derived: launch = %.4g shader cycles -> effective clock %.2f GHz (this trace's duration)" % (cyc, cyc / us / 1e3))
        print("derived: matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 256 CUs x 4 SIMDs) = %.1f %%"
              % (100.0 * tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)))
        print("derived: MFMA work per launch = %.4g x 32 768 flop (busy cycles / 32: a v_mfma_f32_32x32x16_bf16 is 32 busy cycles, a v_mfma_f32_16x16x32_bf16 16 -- twice as many instructions)" % (tot["SQ_VALU_MFMA_BUSY_CYCLES"] / 32))
    if "FETCH_SIZE" in tot:
        print("derived: HBM traffic per launch = %.1f MB read (FETCH_SIZE KiB x 2: gfx950 correction for wide coalesced reads), %.1f MB written"
              % (tot["FETCH_SIZE"] * 1024 * 2 / 1e6, tot.get("WRITE_SIZE", 0) * 1024 / 1e6))
