cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/parity_errors.log
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/gpu_tests_a.log 2>&1; tail -3 gpurun_out/gpu_tests_a.log
timeout -k 10 300 python -u scripts/quick_bf16.py 2>&1 | grep -v amdgpu.ids | tail -3
