cd $GRAFT_REPO_ROOT
for l in slack; do
echo "== $l"; GRLHIP_LIB=$PWD/generativerl_preview_amd/lib_$l.so timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -p no:cacheprovider -k "bf16_ode_sampling" 2>&1 | tail -1
done
timeout -k 10 400 python -u scripts/xtime.py generativerl_preview_amd/lib_base.so generativerl_preview_amd/lib_slack.so generativerl_preview_amd/lib_sp1.so generativerl_preview_amd/lib_sp3.so generativerl_preview_amd/lib_base.so 2>&1 | grep -v amdgpu.ids
