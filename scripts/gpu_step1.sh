cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/parity_errors.log
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/gpu_tests_a.log 2>&1 || { tail -30 gpurun_out/gpu_tests_a.log; exit 1; }
tail -3 gpurun_out/gpu_tests_a.log
bash scripts/profile_bench.sh > gpurun_out/profile_bench.log 2>&1 || { tail -30 gpurun_out/profile_bench.log; exit 1; }
tail -c 1500 gpurun_out/prof/trace.log
