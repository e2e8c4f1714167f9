cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -p no:cacheprovider 2>&1 | tail -2
timeout -k 10 300 python -u scripts/bench_train.py --steps 20 --warmup 3 2>&1 | tail -1 | cut -c100-330
timeout -k 10 300 python -u scripts/bench_train.py --steps 20 --warmup 3 --batch 4096 2>&1 | tail -1 | cut -c100-330
