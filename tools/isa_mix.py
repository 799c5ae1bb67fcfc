#!/usr/bin/env python
"""Instruction mix of one kernel in a hipcc -S --cuda-device-only listing: tools/isa_mix.py file.s <symbol substring>."""
import sys
from collections import Counter
s = open(sys.argv[1]).read()
sym = [l.split(':')[0] + ':' for l in s.split('\n') if ':' in l and sys.argv[2] in l.split(':')[0] and not l.startswith(('.', '\t', ' '))][0]
i = s.index('\n' + sym) + 1
body = s[i:s.index('.Lfunc_end', i)]
lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith((';', '.')) and not l.strip().endswith(':')]
print(sym, 'total instrs', len(lines))
c = Counter(l.split()[0] for l in lines)
print('  '.join('%s %d' % kv for kv in c.most_common(45)))
