#!/usr/bin/env python
"""Per CALL of the funnel: time, HBM traffic and matrix-pipe busy cycles of its kernels, from the CSVs of tools/pmc_funnel.sh (calls per run: counted -- one uprep5_kernel launch per call)."""
import csv, glob, sys, collections
root = sys.argv[1]
KER = ("sweep7_kernel", "expand7_kernel", "threshold7_kernel", "maxthr7_kernel", "resolve7_kernel", "uprep5_kernel", "hist_bloom7_kernel", "sweep4_kernel", "warm4_kernel", "fail_")
def name(k):
    for x in KER:
        if x in k:
            return x
    return None
dur = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in sorted(glob.glob(root + "/stats/**/*_kernel_trace.csv", recursive=True))[:1]:
    for r in csv.DictReader(open(f)):
        n = name(r["Kernel_Name"])
        if n:
            dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            cnt[n] += 1
CALLS = float(max(1, cnt.get("uprep5_kernel", 1)))
ctr = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob(root + "/p*/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        n = name(r["Kernel_Name"])
        if n:
            ctr[n][r["Counter_Name"]] += float(r["Counter_Value"])
print("# per call of pda_score_topk7_f32 (config 3, 262 144 users, raw head); FETCH_SIZE / WRITE_SIZE in KiB summed over the 8 XCDs, FETCH_SIZE x 2 = bytes read")
print("# (gfx950: the counter under-reports wide coalesced reads by 2 x -- guides/MI355X_MICROARCH.md; scattered 16-byte gathers are NOT under-reported: the x 2 figure is an upper bound there)")
tot_us = tot_r = tot_w = 0.0
for n in KER:
    if n not in dur:
        continue
    us = dur[n] / CALLS
    fr = ctr[n].get("FETCH_SIZE", 0.0) * 1024 / CALLS
    wr = ctr[n].get("WRITE_SIZE", 0.0) * 1024 / CALLS
    mf = ctr[n].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / CALLS
    print("%-20s launches %5.1f  %9.1f us  FETCH_SIZE %8.1f MB (x 2: %8.1f MB)  WRITE_SIZE %8.1f MB  MFMA busy cycles %.4g" % (n, cnt[n] / CALLS, us, fr / 1e6, 2 * fr / 1e6, wr / 1e6, mf))
    tot_us += us; tot_r += fr; tot_w += wr
print("%-20s                %9.1f us  FETCH_SIZE %8.1f MB (x 2: %8.1f MB)  WRITE_SIZE %8.1f MB" % ("whole call", tot_us, tot_r / 1e6, 2 * tot_r / 1e6, tot_w / 1e6))
