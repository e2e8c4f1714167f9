cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_dpm_solver.py tests/test_with_grad.py -m gpu -x -q --timeout 300 --timeout-method thread -p no:cacheprovider 2>&1 | tail -15
