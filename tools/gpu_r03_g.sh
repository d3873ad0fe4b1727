R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03g}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 200 python -m pytest tests/test_ops_parity.py -m gpu -x -q -k "lowres or pointwise" > $O/pytest_ops.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_ops.txt; tail -n 2 $O/pytest_ops.txt
grep -q "rc=0" $O/pytest_ops.txt || exit 1
timeout 200 python tools/bench_pw.py 2>&1 | grep -v amdgpu.ids > $O/bench_pw.txt; tail -n 22 $O/bench_pw.txt | cut -c1-110
timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --train-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"
