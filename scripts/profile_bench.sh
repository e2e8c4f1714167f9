#!/bin/bash
# rocprofv3 passes for the bench kernel (run on the GPU box from the repo root):
#   1. kernel trace + stats of the default bench command
#   2-4. PMC passes (own runs, small workload): MFMA/LDS counters, FETCH_SIZE, WRITE_SIZE
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- python3 "$R/bench.py" --no-cpu-baseline > "$OUT/trace.log" 2>&1
SMALL="--states 131072 --steps 1 --warmup 0 --no-cpu-baseline"
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d "$OUT/pmc_sq" -o pmc -- python3 "$R/bench.py" $SMALL > "$OUT/pmc_sq.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o pmc -- python3 "$R/bench.py" $SMALL > "$OUT/pmc_fetch.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d "$OUT/pmc_write" -o pmc -- python3 "$R/bench.py" $SMALL > "$OUT/pmc_write.log" 2>&1
find "$OUT" -name "*.csv" | head -30
