# round 3: first GPU run of the F(4x4,3x3) Winograd schedule: parity tests, then timing against F(2x2) on the batched shapes
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03m}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_ops_parity.py -m gpu -x -q -k "winograd" > $O/pytest_wino.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest_wino.txt
timeout 200 python tools/bench_wino4.py > $O/bench_wino4.txt 2>&1; echo "bench rc=$?"; cat $O/bench_wino4.txt
