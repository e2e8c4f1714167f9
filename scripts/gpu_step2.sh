cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -p no:cacheprovider 2>&1 | tail -4
timeout -k 10 500 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>gpurun_out/bench.err | cut -c1-1500
