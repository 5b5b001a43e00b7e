#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of vio_kernels.hip as the compiler reports it (-Rpass-analysis=kernel-resource-usage).

  python tools/kernel_resources.py [extra hipcc flags, e.g. -DLIN_THREADS=768]   -> a table on stdout
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "visual-inertial-odometry_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-c", "vio_kernels.hip", "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
out = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: (.*) \[-Rpass", line) or re.search(r"remark: (.*) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
keys = ["VGPRs", "AGPRs", "VGPRs Spill", "ScratchSize [bytes/lane]", "TotalSGPRs", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"]
print("%-34s %6s %6s %6s %8s %6s %5s %8s" % ("kernel", "VGPR", "AGPR", "spill", "scratch", "SGPR", "occ", "LDS(st)"))
for r in rows:
    name = re.sub(r"^_Z\d+", "", r["name"])
    name = re.sub(r"(12DeviceTables|9BatchArgs|13ReduceTables|9TriTables).*$", "", name)
    print("%-34s %6s %6s %6s %8s %6s %5s %8s" % tuple([name[:34]] + [r.get(k, "-") for k in keys]))
