#!/usr/bin/env python3
"""Summaries of rocprofv3's rocpd (sqlite) output, for the files committed under profiles/.

  python tools/rocpd_summary.py stats <results.db>          kernel-trace: calls, average / min / max duration per kernel (CSV)
  python tools/rocpd_summary.py pmc <results.db> [...]      counter passes: average counter value per launch per kernel (CSV)
"""
import sqlite3
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0]


def stats(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, duration, grid_x, grid_y from kernels").fetchall()
    agg = defaultdict(list)
    for name, dur, gx, gy in rows:
        agg[short(name)].append(dur)
    tot = sum(sum(v) for v in agg.values())
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print('"%s",%d,%d,%.1f,%.2f,%d,%d' % (k, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot, min(v), max(v)))


def pmc(paths):
    acc = defaultdict(lambda: defaultdict(list))
    for path in paths:
        db = sqlite3.connect(path)
        cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
        ni, ci, vi = cols.index("kernel_name") if "kernel_name" in cols else cols.index("name"), cols.index("counter_name"), cols.index("value")
        for row in db.execute("select * from counters_collection"):
            acc[short(row[ni])][row[ci]].append(float(row[vi]))
    names = sorted({c for k in acc for c in acc[k]})
    print("kernel,launches," + ",".join(names))
    for k in sorted(acc):
        n = max(len(v) for v in acc[k].values())
        print("%s,%d," % (k, n) + ",".join("%.1f" % (sum(acc[k][c]) / max(len(acc[k][c]), 1)) if c in acc[k] else "" for c in names))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
