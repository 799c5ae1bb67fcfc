#!/usr/bin/env python
"""Looks for compiler-inserted `s_waitcnt vmcnt(0)` in front of LDS reads in the gfx950 ISA of a .hip file — the pattern that
serialised the LDS-DMA stream of the transposing-read kernels (late round 3).

  cd lsps_amd/csrc && hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only c8.hip -o /tmp/c8.s
  python tools/scan_dma_waits.py /tmp/c8.s

A wait is "the kernel's own" when an s_barrier follows within three lines (the chunk-end synchronisation); it is reported as
suspicious when the next memory / matrix instruction after it is a ds_read and no barrier follows."""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
fn = None
res = {}
for i, l in enumerate(lines):
    m = re.match(r'^(_ZN4lsps\w+):', l)
    if m:
        fn = m.group(1)
    if fn and 's_waitcnt vmcnt(0)' in l:
        mine = 's_barrier' in ' '.join(lines[i + 1:i + 4])
        j = i + 1
        while j < len(lines) and not re.search(r'ds_read|ds_write|buffer_|global_|s_barrier|v_mfma', lines[j]):
            j += 1
        res.setdefault(fn, []).append((i, mine, lines[j].strip()[:50] if j < len(lines) else ''))
for k, v in res.items():
    sus = [x for x in v if not x[1] and 'ds_read' in x[2]]
    print('%-70s vmcnt(0) waits %3d   before a ds_read without a barrier: %d' % (k[:70], len(v), len(sus)))
    for x in sus[:4]:
        print('      line %d: %s' % (x[0], x[2]))
