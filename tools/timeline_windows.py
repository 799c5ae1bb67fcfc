#!/usr/bin/env python
"""Compresses a tools/timeline.py --all listing into windows: per queue the busy share and the top kernels."""
import collections
import sys
lines = open(sys.argv[1]).read().splitlines()
i0 = [i for i, l in enumerate(lines) if l.strip().startswith('start_us')][0]
ev = []
for l in lines[i0 + 1:]:
    a, b, q, k = l.split(None, 3)
    ev.append((float(a), float(b), q, k.strip()))
w = int(sys.argv[2]) if len(sys.argv) > 2 else 250
T = max(a + b for a, b, _, _ in ev)
for t0 in range(0, int(T) + w, w):
    per = collections.defaultdict(collections.Counter)
    for a, b, q, k in ev:
        ov = min(a + b, t0 + w) - max(a, t0)
        if ov > 0:
            per[q][k.split('<')[0][:22]] += ov
    s = []
    for q in sorted(per):
        tot = sum(per[q].values())
        s.append("%s: %3d%% %s" % (q, 100 * tot / w, ", ".join("%s %d" % (k, v) for k, v in per[q].most_common(2))))
    print("%5d  %s" % (t0, " | ".join(s)))
