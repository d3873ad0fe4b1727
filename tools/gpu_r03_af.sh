# final build, default flags: the small goldens + the conv / norm op tests once more after the lfdm_conv_params layout change (ABI 6)
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03af}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 75 python -m pytest tests/test_golden_gpu.py tests/test_host_and_abi.py -m gpu -x -q -k "not c2" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 2 $O/pytest.txt
