R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03y}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 200 python -m pytest tests/test_ops_parity.py -m gpu -x -q -k "winograd4" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.txt
