cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -p no:cacheprovider 2>&1 | tail -2
bash scripts/profile_bench.sh > gpurun_out/profile_bench.log 2>&1 || { tail -30 gpurun_out/profile_bench.log; exit 1; }
grep -h '^{' gpurun_out/prof/trace.log | cut -c1-200
