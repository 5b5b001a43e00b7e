#!/usr/bin/env python3
"""Summaries of rocprofv3's rocpd (sqlite) output, for the files committed under profiles/.

  python tools/rocpd_summary.py stats <results.db>          kernel-trace: calls, average / min / max duration per kernel (CSV)
  python tools/rocpd_summary.py pmc <results.db> [...]      counter passes: average counter value per launch per kernel (CSV)
  python tools/rocpd_summary.py traffic <out.csv> <out.json> <fetch.db> <write.db>
        HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the two counters come from separate passes, and on gfx950
        FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM section); what bench.py reports as
        roofline.traffic
"""
import json
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


def counter_avg(path, counter):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
    ni, ci, vi = cols.index("kernel_name") if "kernel_name" in cols else cols.index("name"), cols.index("counter_name"), cols.index("value")
    acc = defaultdict(list)
    for row in db.execute("select * from counters_collection"):
        if row[ci] == counter:
            acc[short(row[ni])].append(float(row[vi]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def traffic(out_csv, out_json, fetch_db, write_db):
    fetch, write = counter_avg(fetch_db, "FETCH_SIZE"), counter_avg(write_db, "WRITE_SIZE")
    out = {}
    with open(out_csv, "w") as f:
        f.write("kernel,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,hbm_bytes_per_launch_corrected\n")
        for k in sorted(set(fetch) | set(write)):
            if not k.startswith("k_"):
                continue
            fk, wk = fetch.get(k, 0.0), write.get(k, 0.0)
            hb = (2.0 * fk + wk) * 1024.0
            out[k] = {"fetch_kb": round(fk, 1), "write_kb": round(wk, 1), "hbm_bytes_per_launch": hb}
            f.write("%s,%.3f,%.3f,%.0f\n" % (k, fk, wk, hb))
    json.dump(out, open(out_json, "w"), indent=1)
    print(open(out_csv).read())


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "traffic":
        traffic(*sys.argv[2:6])
    else:
        pmc(sys.argv[2:])
