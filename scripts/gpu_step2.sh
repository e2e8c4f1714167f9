cd $GRAFT_REPO_ROOT
R=$PWD
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_mb; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 60 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/a -o pmc -- $R/scripts/microbench/gemm_x6_bench_3 > $OUT/log.txt 2>&1 || tail -5 $OUT/log.txt
timeout -s KILL 60 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA -d $OUT/b -o pmc -- $R/scripts/microbench/gemm_x6_bench_3 > $OUT/log2.txt 2>&1 || tail -5 $OUT/log2.txt
timeout -s KILL 60 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $OUT/c -o pmc -- $R/scripts/microbench/gemm_x6_bench_3 > $OUT/log3.txt 2>&1 || tail -5 $OUT/log3.txt
timeout -s KILL 60 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/d -o pmc -- $R/scripts/microbench/gemm_x6_bench_3 > $OUT/log4.txt 2>&1 || tail -5 $OUT/log4.txt
python3 - $OUT <<'PY'
import sqlite3, sys, glob
v = {}
for db in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    for r in c.execute("select counter_name, sum(value), count(distinct dispatch_id) from counters_collection where kernel_name like '%gemm_x6%' group by counter_name"):
        v[r[0]] = (r[1], r[2])
for k, x in sorted(v.items()):
    print("  %-34s %.5g  (%d dispatches)" % (k, x[0], x[1]))
PY
