R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03o}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 200 python -m pytest tests/test_ops_parity.py -m gpu -x -q -k "winograd4" > $O/pytest_wino.txt 2>&1; echo "pytest rc=$?"; tail -n 2 $O/pytest_wino.txt
timeout 200 python tools/bench_wino4.py > $O/bench_wino4.txt 2>&1; echo "bench rc=$?"; cat $O/bench_wino4.txt
