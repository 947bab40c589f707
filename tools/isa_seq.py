#!/usr/bin/env python
"""Print the instruction-class sequence around the MFMA chain of one kernel in a -save-temps .s file."""
import sys
s = open(sys.argv[1]).read()
name = sys.argv[2]
i = s.index(name + ':')
k = s[i:s.index('.end_amdhsa_kernel', i)]
lines = k.split('\n')
idx = [n for n, l in enumerate(lines) if 'v_mfma' in l]
seq = []
for l in lines[idx[0] - int(sys.argv[3]) if len(sys.argv) > 3 else idx[0] - 30: idx[-1] + 40]:
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.'):
        continue
    op = l.split()[0]
    if op.startswith('s_waitcnt'):
        seq.append('[' + l.split(None, 1)[1].replace('cnt', '') + ']')
    else:
        seq.append('M' if op.startswith('v_mfma') else 'v' if op.startswith('v_') else 'B' if op.startswith('s_barrier') else 'J' if 'branch' in op else 's' if op.startswith('s_') else 'd' if op.startswith('ds_') else 'g' if op.startswith('global_') else 'x' if op.startswith('scratch_') else '?')
print(''.join(seq))
