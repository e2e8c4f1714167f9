cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/parity_errors.log
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/gpu_tests_a.log 2>&1; tail -5 gpurun_out/gpu_tests_a.log
timeout -k 10 300 python -u scripts/bench_train.py --steps 20 --warmup 3 2>&1 | tail -2 | cut -c1-400
timeout -k 10 500 python -u bench.py --states 131072 --steps 2 --warmup 1 > gpurun_out/bench_small.json 2> gpurun_out/bench_small.err; tail -c 3000 gpurun_out/bench_small.json; tail -3 gpurun_out/bench_small.err
