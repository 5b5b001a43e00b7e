#!/bin/bash
# Diagnostic builds of the library under visual-inertial-odometry_amd/csrc/diag/ (git-ignored, never shipped or timed as the product):
#   tools/build_diag.sh stamps                      -> diag/libvio_hip_stamps.so   (-DVIO_STAMPS: tools/diag_stamps.py, diag_batch_stamps.py)
#   tools/build_diag.sh <name> [-D...]              -> diag/libvio_hip_<name>.so   (e.g. `t768 -DLIN_THREADS=768`; A/B with tools/ab.py)
#   tools/build_diag.sh prev                        -> diag/libvio_hip_prev.so     (a copy of the current product build, as the A/B baseline)
cd "$(dirname "$0")/../visual-inertial-odometry_amd/csrc" || exit 1
mkdir -p diag
name=${1:-stamps}; shift
if [ "$name" = prev ]; then cp libvio_hip.so diag/libvio_hip_prev.so; exit $?; fi
flags="$*"
[ "$name" = stamps ] && flags="-DVIO_STAMPS $flags"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value $flags vio_kernels.hip vio_api.cpp host_dense.cpp vio_plan.cpp -o diag/libvio_hip_$name.so -ldl -lpthread
