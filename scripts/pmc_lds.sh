#!/bin/bash
# LDS-side counters of one sampling call for a given library build: scripts/pmc_lds.sh <name> [lib.so]
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1
[ $# -ge 2 ] && export GRLHIP_LIB=$R/$2
OUT=$R/gpurun_out/pmclds_$NAME
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d "$OUT" -o pmc -- python3 "$R/scripts/one_call.py" > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
python3 - "$OUT" <<'PY'
import sqlite3, sys, glob
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
v = {r[0]: r[1] for r in c.execute("select counter_name, sum(value) from counters_collection where kernel_name like '%k_sample%' group by counter_name")}
for k, x in sorted(v.items()):
    print("%-28s %.4g" % (k, x))
PY
