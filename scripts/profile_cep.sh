#!/bin/bash
# rocprofv3 kernel trace of the QGPO CEP energy-guidance training step (run on the GPU box from the repo root)
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_cep
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o cep -- python3 "$R/scripts/bench_cep.py" --states 65536 --steps 10 > "$OUT/trace.log" 2>&1
python3 - <<PY
import sqlite3, json
c = sqlite3.connect("$OUT/trace/cep_results.db")
rows = [dict(name=r[0][:110], calls=r[1], total_us=r[2], avg_us=r[3], pct=r[4]) for r in c.execute("select * from top_kernels limit 16")]
json.dump(rows, open("$OUT/top_kernels.json", "w"), indent=1)
for r in rows: print("%6d %10.0f %9.1f %5.1f%% %s" % (r["calls"], r["total_us"], r["avg_us"], r["pct"], r["name"]))
# one chunk of the library's step, launch by launch (between two k_bt_inputs launches of the last hip step)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
ks = list(c.execute("select name, start, end%s from kernels order by start" % ((", " + gx) if gx else "")))
marks = [i for i, r in enumerate(ks) if "k_bt_inputs" in r[0]]
a, b = marks[-2], marks[-1]
seq = [dict(k=r[0][:60], us=(r[2] - r[1]) / 1e3, gap_us=(ks[a + i + 1][1] - r[2]) / 1e3, grid=(r[3] if gx else None)) for i, r in enumerate(ks[a:b])]
json.dump(seq, open("$OUT/one_chunk.json", "w"), indent=1)
for q in seq: print("%8.1f %6.1f %s %s" % (q["us"], q["gap_us"], q["grid"], q["k"]))
print("chunk busy us", sum(q["us"] for q in seq), "span us", (ks[b][1] - ks[a][1]) / 1e3)
PY
