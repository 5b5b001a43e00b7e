#!/usr/bin/env python3
"""A/B of two builds of the library on any of the timing diagnostics: alternates subprocesses of `script args...` between
csrc/libvio_hip.so and another .so (VIO_HIP_LIB) and prints every run's best "<x> us per ..." figure and the medians.
  python tools/ab.py <other.so> <rounds> tools/diag_gn_timing.py 20000 2000"""
import os
import re
import statistics
import subprocess
import sys

other = os.path.abspath(sys.argv[1])
rounds = int(sys.argv[2])
cmd = [sys.executable] + sys.argv[3:]
res = {"base": [], "other": []}
for r in range(rounds):
    for name in ("base", "other"):
        env = dict(os.environ)
        if name == "other":
            env["VIO_HIP_LIB"] = other
        out = subprocess.run(cmd, capture_output=True, text=True, env=env).stdout
        us = [float(m) for m in re.findall(r": ([0-9.]+) us per", out)]
        res[name].append(min(us) if us else float("nan"))
        print(r, name, us, flush=True)
for name in ("base", "other"):
    print(name, "median of the runs' best: %.3f us" % statistics.median(res[name]), "min %.3f" % min(res[name]))
