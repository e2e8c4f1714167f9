#!/bin/bash
# rocprofv3 kernel trace of the C3 training step (run on the GPU box from the repo root): per-kernel totals and the
# GPU-idle share of a step (gaps between consecutive dispatches).  Extra arguments go to bench_train.py
# (e.g. --backward-split 2)
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_train
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o train -- python3 "$R/scripts/bench_train.py" --steps 6 --warmup 2 "$@" > "$OUT/trace.log" 2>&1
python3 "$R/scripts/summarize_train_prof.py" "$OUT"
