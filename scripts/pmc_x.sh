#!/bin/bash
# SQ counter breakdown of one sampling call for several library builds: scripts/pmc_x.sh lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  OUT=$R/gpurun_out/pmc_$name; rm -rf $OUT; mkdir -p $OUT
  export GRLHIP_LIB=$R/$lib
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $OUT -o pmc -- python3 $R/scripts/one_call.py > $OUT/log.txt 2>&1 || { echo "pmc run failed for $name"; tail -5 $OUT/log.txt; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/b -o pmc -- python3 $R/scripts/one_call.py > $OUT/log2.txt 2>&1 || { echo "pmc run 2 failed for $name"; tail -5 $OUT/log2.txt; }
  python3 - $OUT $name <<'PY'
import sqlite3, sys, glob
v = {}
for db in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    try:
        for r in c.execute("select counter_name, sum(value) from counters_collection where kernel_name like '%k_sample%' group by counter_name"):
            v[r[0]] = r[1]
    except Exception as e:
        print("db error", e)
wc = v.get("SQ_WAVE_CYCLES", 1)
print("==", sys.argv[2])
for k, x in sorted(v.items()):
    print("  %-28s %.4g  (%.1f%% of wave cycles)" % (k, x, 100 * x / wc / (4 if k == "SQ_VALU_MFMA_BUSY_CYCLES" else 1)))
PY
done
