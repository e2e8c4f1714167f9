cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -p no:cacheprovider 2>&1 | tail -2
timeout -k 10 300 python -u scripts/quick_c4.py > gpurun_out/c4.log 2>&1; grep -E "C5" gpurun_out/c4.log | cut -c1-260
