#!/usr/bin/env python3
"""Fixture for the trajectory metric of the stream configuration: the reference's shipped
simulation trajectories (TUM files under assignments/17-vins-initialization/doc/with-noise/comparison/) together with
the `evo_ape tum ... -va` statistics it published for them (summary.csv in the same directory).  Data only; run in the
build container where /root/reference is mounted:  python tests/golden/make_golden_ape.py"""
import csv
import os

import numpy as np

SRC = "/root/reference/workspace/assignments/17-vins-initialization/doc/with-noise/comparison"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ape_reference.npz")


def tum(path):
    return np.array([[float(v) for v in line.split()[:8]] for line in open(path) if line.strip()], dtype=np.float64)


def main():
    out = {"ground_truth": tum(os.path.join(SRC, "ground-truth.txt"))}
    with open(os.path.join(SRC, "summary.csv")) as f:
        rows = list(csv.reader(f))
    keys = rows[0][1:]
    for r in rows[1:]:
        name = r[0][:-4].replace("-", "_")
        out["est_" + name] = tum(os.path.join(SRC, r[0]))
        for k, v in zip(keys, r[1:]):
            out["stat_%s_%s" % (name, k)] = float(v)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
