#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_tcc; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $OUT/a -o pmc -- python3 $R/scripts/one_call.py > $OUT/log.txt 2>&1 || tail -5 $OUT/log.txt
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/b -o pmc -- python3 $R/scripts/one_call.py > $OUT/log2.txt 2>&1 || tail -5 $OUT/log2.txt
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/c -o pmc -- python3 $R/scripts/one_call.py > $OUT/log3.txt 2>&1 || tail -5 $OUT/log3.txt
python3 - $OUT <<'PY'
import sqlite3, sys, glob
v = {}
for db in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    try:
        for r in c.execute("select counter_name, sum(value) from counters_collection where kernel_name like '%k_sample%' group by counter_name"):
            v[r[0]] = r[1]
    except Exception as e:
        print("db error", e)
for k, x in sorted(v.items()):
    print("  %-34s %.5g" % (k, x))
PY
