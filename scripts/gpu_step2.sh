cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -p no:cacheprovider -k "bf16" 2>&1 | tail -3
timeout -k 10 400 python -u scripts/xtime.py generativerl_preview_amd/lib_base.so generativerl_preview_amd/libgrlhip.so generativerl_preview_amd/lib_base.so generativerl_preview_amd/libgrlhip.so 2>&1 | grep -v amdgpu.ids
