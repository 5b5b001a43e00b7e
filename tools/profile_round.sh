#!/bin/bash
# The round's profile set, on the GPU box:   tools/profile_round.sh <tag>   -> gpurun_out/<tag>/  (copy what is judged into profiles/)
#   kernel stats (rocprofv3 --kernel-trace --stats) of the default bench command for both landmark kinds,
#   HBM traffic (separate --pmc FETCH_SIZE / WRITE_SIZE passes over tools/diag_gn_loop.py), SQ counters of the inverse-depth loop,
#   the bench lines themselves (default flags, xyz, two ranks on the one device) without the profiler.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
T=${1:-r03}
O=gpurun_out/$T
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-per-frame > $O/${T}_bench_under_rocprof.log 2>$O/err1.log
rocprofv3 --kernel-trace --stats -d $O/stats_xyz -o s -- python3 bench.py --landmark-type xyz --steps 200 --warmup 20 --no-cpu-baseline --no-per-frame > $O/${T}_bench_xyz_under_rocprof.log 2>$O/err2.log
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o f -- python3 tools/diag_gn_loop.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o w -- python3 tools/diag_gn_loop.py > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch_xyz -o f -- python3 tools/diag_gn_loop.py xyz > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write_xyz -o w -- python3 tools/diag_gn_loop.py xyz > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $O/sq -o sq -- python3 tools/diag_gn_loop.py > /dev/null 2>&1
db() { find $O/$1 -name "*.db" | head -1; }
python3 tools/rocpd_summary.py stats $(db stats) > $O/${T}_kernel_stats.csv
python3 tools/rocpd_summary.py stats $(db stats_xyz) > $O/${T}_kernel_stats_xyz.csv
python3 tools/rocpd_summary.py traffic $O/${T}_pmc_traffic.csv $O/traffic.json $(db fetch) $(db write)
python3 tools/rocpd_summary.py traffic $O/${T}_xyz_pmc_traffic.csv $O/traffic_xyz.json $(db fetch_xyz) $(db write_xyz)
python3 tools/rocpd_summary.py pmc $(db sq) > $O/${T}_sq_counters.csv
rm -rf $O/stats $O/stats_xyz $O/fetch $O/write $O/fetch_xyz $O/write_xyz $O/sq
python3 bench.py > $O/${T}_bench.json 2> $O/bench_err.log
python3 bench.py --landmark-type xyz > $O/${T}_bench_xyz.json 2> $O/bench_xyz_err.log
VIO_BENCH_ONE_DEVICE=1 python3 bench.py --gpus 2 --steps 50 --warmup 10 > $O/${T}_bench_2ranks_one_device.json 2> $O/bench2_err.log
head -8 $O/${T}_kernel_stats.csv; head -6 $O/${T}_kernel_stats_xyz.csv; cat $O/${T}_pmc_traffic.csv | head -12
tail -c 1500 $O/${T}_bench.json
