#!/bin/bash
# Per-kernel durations (rocprofv3 --kernel-trace) of the GN loop at 20k / 200k landmarks and of the batched loop.
# Usage (on the GPU box): tools/prof_kernels.sh <tag>      -> gpurun_out/<tag>/kernels_{20k,200k,batched}.csv
# VIO_HIP_LIB selects another build of the library.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/$1
mkdir -p $O
rocprofv3 --kernel-trace -d $O/t20k -o s -- python3 tools/diag_gn_timing.py 20000 300 > $O/run_20k.log 2>&1
rocprofv3 --kernel-trace -d $O/t200k -o s -- python3 tools/diag_gn_timing.py 200000 100 > $O/run_200k.log 2>&1
rocprofv3 --kernel-trace -d $O/tb -o s -- python3 tools/diag_batch_gn_timing.py 64 20000 10 > $O/run_batched.log 2>&1
for t in 20k 200k; do python3 tools/rocpd_summary.py stats $(find $O/t$t -name "*.db" | head -1) > $O/kernels_$t.csv; done
python3 tools/rocpd_summary.py stats $(find $O/tb -name "*.db" | head -1) > $O/kernels_batched.csv
rm -rf $O/t20k $O/t200k $O/tb
for t in 20k 200k batched; do echo "== $t"; head -7 $O/kernels_$t.csv | cut -d, -f1,2,4,5; done
