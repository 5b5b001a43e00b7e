#!/usr/bin/env python3
"""Adds what bench.py's batched.roofline reads to traffic_batched.json: the window count of the counter passes and the fp64 operations
per launch from the SQ_INSTS_VALU_* pass (tools/profile_batched.sh).
  python tools/batched_counters_json.py <traffic_batched.json> <batched_mfma.csv> <windows>"""
import csv
import json
import sys

path, mfma_csv, windows = sys.argv[1], sys.argv[2], int(sys.argv[3])
tr = json.load(open(path))
fp = {}
for r in csv.DictReader(open(mfma_csv)):
    mfma = float(r["SQ_INSTS_VALU_MFMA_MOPS_F64"]) * 512
    valu = (2 * float(r["SQ_INSTS_VALU_FMA_F64"]) + float(r["SQ_INSTS_VALU_MUL_F64"]) + float(r["SQ_INSTS_VALU_ADD_F64"])) * 64
    if mfma + valu > 0:
        fp[r["kernel"]] = {"mfma_flops_per_launch": mfma, "valu_flops_per_launch": valu, "flops_per_launch": mfma + valu}
tr["_windows"] = windows
tr["_fp64"] = fp
tr["_note"] = ("%d windows of 20 000 landmarks (tools/profile_batched.sh). hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE "
               "(MI355X_MICROARCH.md: gfx950 FETCH_SIZE counts half of a coalesced stream). _fp64: SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 flops; "
               "VALU fp64 instructions x 64 lanes (an upper bound: exec masks not counted), FMA = 2 flops" % windows)
json.dump(tr, open(path, "w"), indent=1)
