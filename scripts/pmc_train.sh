#!/bin/bash
# PMC counters of the training GEMM kernels (one pass per counter set)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_train; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/a -o pmc -- python3 $R/scripts/bench_train.py --steps 2 --warmup 1 > $OUT/log.txt 2>&1 || tail -5 $OUT/log.txt
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/b -o pmc -- python3 $R/scripts/bench_train.py --steps 2 --warmup 1 > $OUT/log2.txt 2>&1 || tail -5 $OUT/log2.txt
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/c -o pmc -- python3 $R/scripts/bench_train.py --steps 2 --warmup 1 > $OUT/log3.txt 2>&1 || tail -5 $OUT/log3.txt
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $OUT/d -o pmc -- python3 $R/scripts/bench_train.py --steps 2 --warmup 1 > $OUT/log4.txt 2>&1 || tail -5 $OUT/log4.txt
python3 - $OUT <<'PY'
import sqlite3, sys, glob, collections
v = collections.defaultdict(dict)
for db in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    try:
        for r in c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection where kernel_name like '%k_bt_%' group by kernel_name, counter_name"):
            k = r[0].split("(")[0][-40:]
            v[k][r[1]] = r[2]; v[k]["n"] = r[3]
    except Exception as e:
        print("db error", e)
for k, d in v.items():
    print(k, d.get("n"))
    for name, x in sorted(d.items()):
        if name != "n": print("    %-28s %.5g" % (name, x))
PY
