#!/usr/bin/env python
"""Main-loop structure of an MFMA kernel in a hipcc -S listing: tools/isa_loop.py file.s <symbol substring> [mfmas per instance]
Prints labels / branches / barriers around the first instance's MFMAs, the number of instructions in each MFMA gap, and
any VALU result that an MFMA reads within two instructions (the asm-MFMA operand hazard)."""
import re
import sys

s = open(sys.argv[1]).read()
sym = [l.split(':')[0] for l in s.split('\n') if ':' in l and sys.argv[2] in l.split(':')[0] and not l.startswith(('.', '\t', ' '))][0]
i = s.index('\n' + sym + ':')
body = s[i:s.index('.Lfunc_end', i)]
lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith((';', '.amd', '.p2', '.sec', '.glob', '.type', '.prot'))]
idx = [k for k, l in enumerate(lines) if l.startswith('v_mfma')]
n = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx)
print(sym, 'mfmas', len(idx), 'showing first', n)
for k in range(max(0, idx[0] - 80), idx[n - 1] + 40):
    l = lines[k]
    if l.startswith(('.LBB', 's_cbranch', 's_branch', 's_barrier', 's_endpgm', 's_nop')):
        print('  %d %s' % (k, l[:60]))
print('gaps', [idx[k + 1] - idx[k] - 1 for k in range(n - 1)])
for k in idx[:n]:
    m = re.match(r'v_mfma\S+ \S+, (v\d+), (v\d+),', lines[k])
    if not m:
        continue
    for back in (1, 2):
        pl = lines[k - back]
        w = re.match(r'v_\S+ (v\d+),', pl)
        if w and not pl.startswith('v_mfma') and w.group(1) in m.groups():
            print('HAZARD?', k, pl, '|', lines[k])
waits = [l for l in lines[idx[0]:idx[n - 1]] if l.startswith('s_waitcnt')]
print('waitcnts in the span:', len(waits))
