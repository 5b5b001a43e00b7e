export TMPDIR=/tmp
O=gpurun_out/r02r
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-per-frame > $O/bench_under_rocprof.log 2>$O/err1.log
rocprofv3 --kernel-trace --stats -d $O/stats_xyz -o s -- python3 bench.py --landmark-type xyz --steps 200 --warmup 20 --no-cpu-baseline --no-per-frame > $O/bench_xyz_under_rocprof.log 2>$O/err2.log
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o f -- python3 tools/diag_gn_loop.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o w -- python3 tools/diag_gn_loop.py > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch_xyz -o f -- python3 tools/diag_gn_loop.py xyz > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write_xyz -o w -- python3 tools/diag_gn_loop.py xyz > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $O/sq -o sq -- python3 tools/diag_gn_loop.py xyz > /dev/null 2>&1
find $O -name "*.db" | head -20
python tools/rocpd_summary.py stats $O/stats/s_results.db | head -12
python tools/rocpd_summary.py stats $O/stats_xyz/s_results.db | head -12
