# smallest A/B call: named parity tests + one profiled sampling run (kernel table + step sequence)
# usage: gpu_ab.sh TAG "pytest -k expression"
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-ab}; K=${2:-quantile}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_ops_parity.py tests/test_end_to_end.py -m gpu -x -q -k "$K" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -n 3 $O/pytest.txt
bash tools/prof_sequence.sh $TAG > $O/prof.txt 2>&1; tail -n 1 $O/step_sequence.txt
