#!/bin/bash
# Per-phase attribution of k_linearize_hb's SQ counters under the batched loop (64 windows of 20 000 landmarks, throughput policy):
# builds of the library whose item workgroups leave behind the head / phase 1 / phase 1.5 / phase 2 (-DLIN_EXIT_AFTER=p, tools/build_diag.sh),
# the same counter pass for each and for the full kernel; consecutive differences are the phases.  On the GPU box:
#   tools/profile_phases.sh <tag>   -> gpurun_out/<tag>/<tag>_phase_counters.csv   (the diag libraries must have been built: exit0 .. exit3)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
T=${1:-r05}
B=${2:-64}
O=gpurun_out/$T
mkdir -p $O
for v in exit0 exit1 exit2 exit3 full; do
    if [ $v = full ]; then unset VIO_HIP_LIB; else export VIO_HIP_LIB=$PWD/visual-inertial-odometry_amd/csrc/diag/libvio_hip_$v.so; fi
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $O/p_$v -o sq -- python3 tools/diag_batch_gn_timing.py $B 20000 4 > /dev/null 2>&1
    rocprofv3 --kernel-trace -d $O/k_$v -o s -- python3 tools/diag_batch_gn_timing.py $B 20000 6 > /dev/null 2>&1
    python3 tools/rocpd_summary.py pmc $(find $O/p_$v -name "*.db" | head -1) | grep -E "^kernel|k_linearize_hb" > $O/${T}_phase_${v}_sq.csv
    python3 tools/rocpd_summary.py stats $(find $O/k_$v -name "*.db" | head -1) | grep -E "Name|k_linearize_hb" > $O/${T}_phase_${v}_kernel.csv
    rm -rf $O/p_$v $O/k_$v
done
unset VIO_HIP_LIB
python3 - "$O" "$T" <<'PY'
import csv, sys
O, T = sys.argv[1], sys.argv[2]
rows = {}
for v in ("exit0", "exit1", "exit2", "exit3", "full"):
    r = list(csv.DictReader(open("%s/%s_phase_%s_sq.csv" % (O, T, v))))
    k = list(csv.DictReader(open("%s/%s_phase_%s_kernel.csv" % (O, T, v))))
    d = {c: float(x) for c, x in r[0].items() if c not in ("Name", "name", "kernel", "launches") and x != ""} if r else {}
    d["kernel_us"] = float(k[0].get("AverageNs", 0)) / 1e3 if k else 0.0
    rows[v] = d
names = ["head", "phase 1", "phase 1.5", "phase 2", "combine"]
order = ["exit0", "exit1", "exit2", "exit3", "full"]
cols = sorted(rows["full"].keys())
with open("%s/%s_phase_counters.csv" % (O, T), "w") as f:
    f.write("phase," + ",".join(cols) + "\n")
    prev = {c: 0.0 for c in cols}
    for nm, v in zip(names, order):
        f.write(nm + "," + ",".join("%.6g" % (rows[v].get(c, 0.0) - prev[c]) for c in cols) + "\n")
        prev = {c: rows[v].get(c, 0.0) for c in cols}
    f.write("whole kernel," + ",".join("%.6g" % rows["full"].get(c, 0.0) for c in cols) + "\n")
print(open("%s/%s_phase_counters.csv" % (O, T)).read())
PY
