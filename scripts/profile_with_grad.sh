#!/bin/bash
# rocprofv3 kernel trace of the differentiable-sampling policy-gradient step (run on the GPU box from the repo root)
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_wg
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o wg -- python3 "$R/scripts/bench_with_grad.py" --steps 5 > "$OUT/trace.log" 2>&1
python3 - <<PY
import sqlite3, json
c = sqlite3.connect("$OUT/trace/wg_results.db")
rows = [dict(name=r[0][:110], calls=r[1], total_us=r[2], avg_us=r[3], pct=r[4]) for r in c.execute("select * from top_kernels limit 12")]
json.dump(rows, open("$OUT/top_kernels.json", "w"), indent=1)
for r in rows: print("%6d %10.0f %9.1f %5.1f%% %s" % (r["calls"], r["total_us"], r["avg_us"], r["pct"], r["name"]))
PY
tail -2 "$OUT/trace.log"
