#!/bin/bash
# Build product + diagnostic (stamps) libraries, then run GPU parity tests, the k_pose_solve stamp report and a short
# bench on an MI355X box.  Usage: tools/gpu_cycle.sh [bench-steps]
set -e
cd "$(dirname "$0")/.."
python -c "
import __graft_entry__ as g
g.build_hip(True)" 2>&1 | grep -E "error" && exit 1
( cd visual-inertial-odometry_amd/csrc && mkdir -p diag && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DVIO_STAMPS -o diag/libvio_hip_stamps.so vio_kernels.hip vio_api.cpp host_dense.cpp vio_plan.cpp 2>&1 | grep error ) && exit 1
/usr/local/graft/bin/gpurun --timeout 900 -- 'timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3; python tools/diag_stamps.py 2000 2>&1 | tail -5; python bench.py --steps '${1:-100}' --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"ms_per_step\"], d[\"roofline\"][\"kernel_us\"])"' 2>&1 | grep -v "^\[gpurun\] sending\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl"
