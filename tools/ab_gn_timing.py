#!/usr/bin/env python3
"""A/B of two builds of the library on the GN loop (diagnostic): alternates subprocesses of tools/diag_gn_timing.py between
csrc/libvio_hip.so and another .so, a few rounds each, and prints every run and the medians.
  python tools/ab_gn_timing.py <other.so> [rounds] [landmarks]"""
import os
import re
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
other = os.path.abspath(sys.argv[1])
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = sys.argv[3] if len(sys.argv) > 3 else "20000"
res = {"base": [], "other": []}
for r in range(rounds):
    for name in ("base", "other"):
        env = dict(os.environ)
        if name == "other":
            env["VIO_HIP_LIB"] = other
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "diag_gn_timing.py"), n, "2000"], capture_output=True, text=True, env=env).stdout
        us = [float(m) for m in re.findall(r": ([0-9.]+) us per iteration", out)]
        res[name].append(min(us))
        print(r, name, us)
for name in ("base", "other"):
    print(name, "median of the runs' best: %.3f us" % statistics.median(res[name]), "min %.3f" % min(res[name]))
