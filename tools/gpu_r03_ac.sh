R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03ac}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 200 python -m pytest tests/test_golden_gpu.py -m gpu -x -q -k "generator" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 2 $O/pytest.txt
