#!/bin/bash
# Counter passes under the BATCHED loop (vio_batch_gn_iteration, 64 windows of 20 000 landmarks, throughput item policy), on the GPU box:
#   tools/profile_batched.sh <tag>   -> gpurun_out/<tag>/<tag>_batched_{kernels,pmc_traffic,sq_counters,mfma}.csv, traffic_batched.json
# Separate passes for FETCH_SIZE and WRITE_SIZE (they do not fit one TCC pass), the SQ set, and the fp64 MFMA operation count;
# the program follows `--` directly (no shell, no env wrapper: the profiler's library has initialised the GPU by then).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
T=${1:-r04}
B=${2:-64}
O=gpurun_out/$T
mkdir -p $O
rocprofv3 --kernel-trace -d $O/bk -o s -- python3 tools/diag_batch_gn_timing.py $B 20000 10 > $O/${T}_batched_run.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/bf -o f -- python3 tools/diag_batch_gn_timing.py $B 20000 4 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/bw -o w -- python3 tools/diag_batch_gn_timing.py $B 20000 4 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $O/bsq -o sq -- python3 tools/diag_batch_gn_timing.py $B 20000 4 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 -d $O/bm -o m -- python3 tools/diag_batch_gn_timing.py $B 20000 4 > /dev/null 2>&1
db() { find $O/$1 -name "*.db" | head -1; }
python3 tools/rocpd_summary.py stats $(db bk) > $O/${T}_batched_kernels.csv
python3 tools/rocpd_summary.py traffic $O/${T}_batched_pmc_traffic.csv $O/traffic_batched.json $(db bf) $(db bw)
python3 tools/rocpd_summary.py pmc $(db bsq) > $O/${T}_batched_sq_counters.csv
python3 tools/rocpd_summary.py pmc $(db bm) > $O/${T}_batched_mfma.csv
python3 tools/batched_counters_json.py $O/traffic_batched.json $O/${T}_batched_mfma.csv $B
rm -rf $O/bk $O/bf $O/bw $O/bsq $O/bm
head -8 $O/${T}_batched_kernels.csv; cat $O/${T}_batched_pmc_traffic.csv; cat $O/${T}_batched_sq_counters.csv; cat $O/${T}_batched_mfma.csv; tail -3 $O/${T}_batched_run.log
