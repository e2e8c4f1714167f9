#!/bin/bash
# rocprofv3 passes for the bench kernel (run on the GPU box from the repo root).  Every pass is its own run:
#   1. kernel trace + stats of the default bench command (no CPU leg, no secondaries)
#   2. SQ counters (MFMA busy, wave cycles, waits, LDS) on a small workload
#   3. FETCH_SIZE   4. WRITE_SIZE + GRBM_GUI_ACTIVE
# scripts/summarize_prof.py then sums the counters over the k_sample_bf16 dispatches only and writes
# gpurun_out/prof/r03_bench_pmc.json; copy that to profiles/.
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
FULL="--no-cpu-baseline --no-secondary"
SMALL="--states 131072 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- python3 "$R/bench.py" $FULL > "$OUT/trace.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d "$OUT/pmc_sq" -o pmc -- python3 "$R/bench.py" $SMALL > "$OUT/pmc_sq.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o pmc -- python3 "$R/bench.py" $SMALL > "$OUT/pmc_fetch.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d "$OUT/pmc_write" -o pmc -- python3 "$R/bench.py" $SMALL > "$OUT/pmc_write.log" 2>&1
python3 "$R/scripts/summarize_prof.py" "$OUT" "$OUT/r03_bench_pmc.json" 4194304
