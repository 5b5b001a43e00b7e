#!/usr/bin/env python3
"""Diagnostic: per-kernel durations and the idle gaps between consecutive kernels of one stream, from a rocprofv3
--kernel-trace CSV (…_kernel_trace.csv).  Usage: python tools/kernel_gaps.py <dir> [last N dispatches]"""
import csv
import glob
import sys
from collections import defaultdict

f = (glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv") + glob.glob(sys.argv[1] + "/*_kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
rows = rows[-n:]
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
prev_end = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[name] += e - s
    cnt[name] += 1
    if prev_end is not None:
        gap[name] += s - prev_end
    prev_end = e
tot_d = tot_g = 0.0
for k in sorted(dur, key=lambda k: -dur[k]):
    print("%-16s n=%4d  dur %8.2f us   gap before %6.2f us" % (k, cnt[k], dur[k] / cnt[k] / 1e3, gap[k] / cnt[k] / 1e3))
    tot_d += dur[k]; tot_g += gap[k]
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("span %.1f us: kernels %.1f us, gaps %.1f us" % (span / 1e3, tot_d / 1e3, tot_g / 1e3))
