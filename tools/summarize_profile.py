#!/usr/bin/env python3
"""Turns gpurun_out/ rocprofv3 output into the committed summaries under profiles/.

  python tools/summarize_profile.py <round tag> <kernel-stats dir> [<pmc fetch dir> <pmc write dir>]

Writes profiles/<tag>_kernel_stats.csv (verbatim rocprofv3 --kernel-trace --stats summary), and, when the two PMC
passes are given, profiles/<tag>_pmc_traffic.csv and profiles/traffic.json (what bench.py reports as
roofline.traffic).  HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch: FETCH_SIZE and WRITE_SIZE are
collected in separate passes (they do not fit one TCC pass) and on gfx950 FETCH_SIZE tallies 128-byte requests at
64 bytes, so it is doubled (MI355X_MICROARCH.md, HBM section; calibrated there for wide coalesced streams only —
for the gather-heavy kernels here the corrected figure is an upper bound).
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, stats_dir = sys.argv[1], sys.argv[2]
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
src = (glob.glob(os.path.join(stats_dir, "*", "*_kernel_stats.csv")) + glob.glob(os.path.join(stats_dir, "*_kernel_stats.csv")))[0]
shutil.copy(src, os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
print(open(src).read()[:1500])


def per_kernel(pmc_dir, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(pmc_dir, "*", "*_counter_collection.csv")) + glob.glob(os.path.join(pmc_dir, "*_counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                name = row["Kernel_Name"].split("(")[0]
                tot[name] += float(row["Counter_Value"])
                cnt[name] += 1
    return {k: tot[k] / cnt[k] for k in tot}


if len(sys.argv) >= 5:
    fetch, write = per_kernel(sys.argv[3], "FETCH_SIZE"), per_kernel(sys.argv[4], "WRITE_SIZE")
    out = {}
    with open(os.path.join(ROOT, "profiles", tag + "_pmc_traffic.csv"), "w") as f:
        f.write("kernel,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,hbm_bytes_per_launch_corrected\n")
        for k in sorted(set(fetch) | set(write)):
            if not k.startswith("k_"):
                continue
            fk, wk = fetch.get(k, 0.0), write.get(k, 0.0)
            hb = (2.0 * fk + wk) * 1024.0
            out[k] = {"fetch_kb": fk, "write_kb": wk, "hbm_bytes_per_launch": hb}
            f.write("%s,%.3f,%.3f,%.0f\n" % (k, fk, wk, hb))
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(open(os.path.join(ROOT, "profiles", tag + "_pmc_traffic.csv")).read())
