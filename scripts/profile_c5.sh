#!/bin/bash
# rocprofv3 passes for C5 on the bf16 batch engine (GPU box, repo root): kernel trace + stats of scripts/quick_c5.py, then
# SQ / FETCH_SIZE / WRITE_SIZE counters in runs of their own (5 solver steps); scripts/summarize_c5.py writes the JSON.
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/c5
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o c5 -- python3 "$R/scripts/quick_c5.py" 262144 bf16 > "$OUT/trace.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d "$OUT/pmc_sq" -o pmc -- python3 "$R/scripts/quick_c5.py" 262144 bf16 5 > "$OUT/pmc_sq.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o pmc -- python3 "$R/scripts/quick_c5.py" 262144 bf16 5 > "$OUT/pmc_fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d "$OUT/pmc_write" -o pmc -- python3 "$R/scripts/quick_c5.py" 262144 bf16 5 > "$OUT/pmc_write.log" 2>&1
python3 "$R/scripts/summarize_c5.py" "$OUT" "$OUT/c5_kernels.json"
