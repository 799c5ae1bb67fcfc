#!/usr/bin/env python
"""One step's kernel timeline from a rocprofv3 `--kernel-trace --output-format csv` trace: which launches run on which
queue, when, for how long, and how much of the step each queue / both queues are busy.  A step = the launches between
two consecutive `adam_kernel` dispatches (every update method ends with ONE Adam launch).
Usage: tools/timeline.py <..._kernel_trace.csv> [step index from the end, default 2] [--all]"""
import csv
import sys


def short(name):
    n = name.replace('lsps::', '').replace('void ', '')
    return n.split('(')[0][:48]


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith('--') else 2
    rows = list(csv.DictReader(open(path)))
    if not rows:
        print("empty trace")
        return
    cols = rows[0].keys()
    ks = 'Kernel_Name' if 'Kernel_Name' in cols else [c for c in cols if 'ame' in c][0]
    s0 = 'Start_Timestamp' if 'Start_Timestamp' in cols else [c for c in cols if 'tart' in c][0]
    s1 = 'End_Timestamp' if 'End_Timestamp' in cols else [c for c in cols if c.startswith('End')][0]
    qk = 'Stream_Id' if 'Stream_Id' in cols else 'Queue_Id'
    ev = sorted(((int(r[s0]), int(r[s1]), r.get(qk, '?') + '/' + r.get('Queue_Id', '?'), r[ks]) for r in rows))
    adam = [i for i, e in enumerate(ev) if 'adam_kernel' in e[3]]
    if len(adam) < back + 1:
        print("only %d adam launches" % len(adam))
        return
    a, b = adam[-back - 1], adam[-back]
    step = ev[a + 1:b + 1]
    t0 = ev[a][1]
    print("# columns: %s ; queue key = %s/Queue_Id" % (list(cols), qk))
    print("# step between adam #%d and #%d: %d launches, span %.3f ms (end of previous adam -> end of this adam)"
          % (len(adam) - back - 1, len(adam) - back, len(step), (step[-1][1] - t0) / 1e6))
    queues = sorted(set(e[2] for e in step))
    busy = {}
    for q in queues:
        busy[q] = sum(e[1] - e[0] for e in step if e[2] == q)
    # union of busy intervals over all queues
    iv = sorted((e[0], e[1]) for e in step)
    union, cur0, cur1 = 0, iv[0][0], iv[0][1]
    for x0, x1 in iv[1:]:
        if x0 > cur1:
            union += cur1 - cur0
            cur0, cur1 = x0, x1
        else:
            cur1 = max(cur1, x1)
    union += cur1 - cur0
    total = sum(busy.values())
    print("# queues: " + ", ".join("%s busy %.3f ms (%d launches)" % (q, busy[q] / 1e6, sum(1 for e in step if e[2] == q))
                                   for q in queues))
    print("# sum of kernel durations %.3f ms, union (any queue busy) %.3f ms, overlapped %.3f ms, idle %.3f ms"
          % (total / 1e6, union / 1e6, (total - union) / 1e6, (step[-1][1] - t0 - union) / 1e6))
    agg = {}
    for e in step:
        k = (e[2], short(e[3]))
        c = agg.setdefault(k, [0, 0])
        c[0] += 1
        c[1] += e[1] - e[0]
    print("# per (queue, kernel): calls, total us")
    for (q, n), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("#   %-8s %-48s %4d %9.1f" % (q, n, c, t / 1e3))
    if '--all' in sys.argv:
        print("%10s %9s %-8s %s" % ("start_us", "dur_us", "queue", "kernel"))
        for e in step:
            print("%10.1f %9.1f %-8s %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2], short(e[3])))


if __name__ == '__main__':
    main()
