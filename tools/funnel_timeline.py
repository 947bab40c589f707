"""One funnel call's launches out of a rocprofv3 kernel trace: tools/funnel_timeline.py gpurun_out/prof_x/run_kernel_trace.csv [which call = 2]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
names = ('select7', 'expand7', 'threshold7', 'sweep7', 'resolve7', 'sweep4', 'warm4', 'fail_', 'uprep5', 'init7', 'bloom', 'warm_mask')
sel = sorted([r for r in rows if any(x in r['Kernel_Name'] for x in names)], key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(sel) if 'init7' in r['Kernel_Name']]
w = int(sys.argv[2]) if len(sys.argv) > 2 else 2
a, b = idx[w], idx[w + 1] if w + 1 < len(idx) else len(sel)
t0 = int(sel[a]['Start_Timestamp'])
tot = {}
for r in sel[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0].split('<')[0][:40]
    if 'init7' in r['Kernel_Name'] and s != t0:
        break
    tot[nm] = tot.get(nm, 0) + (e - s)
    print("%9.1f us +%8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, nm))
print({k: round(v / 1e3, 1) for k, v in tot.items()})
