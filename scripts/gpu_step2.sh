cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout -k 10 600 python -u bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 4000 gpurun_out/bench_final.json; tail -3 gpurun_out/bench_final.err
timeout -k 10 120 python -u -m pytest tests/test_zz_no_alloc.py -m gpu -q --timeout 100 --timeout-method thread -p no:cacheprovider 2>&1 | tail -1
