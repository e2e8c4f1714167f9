cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u scripts/quick_bf16.py 2>&1 | grep -v amdgpu.ids | tail -4
