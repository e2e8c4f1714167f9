#!/bin/bash
# LDS / VMEM latency counters of one sampling call: scripts/pmc_lat.sh <name> [lib.so]
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1
[ $# -ge 2 ] && export GRLHIP_LIB=$R/$2
OUT=$R/gpurun_out/pmclat_$NAME
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU -d "$OUT" -o pmc -- python3 "$R/scripts/one_call.py" > "$OUT/log.txt" 2>&1
python3 - "$OUT" <<'PY'
import sqlite3, sys, glob
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
v = {r[0]: r[1] for r in c.execute("select counter_name, sum(value) from counters_collection where kernel_name like '%k_sample%' group by counter_name")}
for k, x in sorted(v.items()):
    print("%-28s %.4g" % (k, x))
print("LDS level/insts %.1f   VMEM level/insts %.1f" % (v["SQ_INST_LEVEL_LDS"] / v["SQ_INSTS_LDS"], v["SQ_INST_LEVEL_VMEM"] / max(1, v["SQ_INSTS_VMEM_RD"])))
PY
