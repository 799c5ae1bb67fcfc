#!/usr/bin/env python
"""Turns rocprofv3 output (rocpd .db from --kernel-trace --stats, or counter_collection.csv from --pmc)
into the small text summaries committed under profiles/."""
import collections
import csv
import sqlite3
import sys


def kernel_stats(db):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    out = ["%-78s %7s %14s %12s %7s" % ("kernel", "calls", "total_ms", "avg_ms", "pct")]
    for name, calls, total, avg, pct in rows:
        out.append("%-78s %7d %14.1f %12.2f %7.2f" % (name[:78], calls, total / 1e3, avg / 1e3, pct))
    return "\n".join(out)


def pmc_stats(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    out = []
    for k, v in agg.items():
        out.append(k)
        for cn, xs in sorted(v.items()):
            out.append("    %-28s mean/dispatch %16.0f   dispatches %d" % (cn, sum(xs) / len(xs), len(xs)))
    return "\n".join(out)


if __name__ == '__main__':
    for p in sys.argv[1:]:
        print("# " + p)
        print(kernel_stats(p) if p.endswith('.db') else pmc_stats(p))
        print()
