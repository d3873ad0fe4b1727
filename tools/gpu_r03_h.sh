R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03h}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 240 python -m pytest tests/test_ops_parity.py tests/test_end_to_end.py -m gpu -x -q -k "planar_in or sampler or quantile or sample or graph" > $O/pytest_ops.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_ops.txt; tail -n 2 $O/pytest_ops.txt
grep -q "rc=0" $O/pytest_ops.txt || exit 1
for cfg in "LFDM_STEM_MFMA=0" "LFDM_STEM_MFMA=1" "LFDM_RES_STREAM=1" "LFDM_RES_STREAM=1 LFDM_RES_STREAM_MAX_ROWS=2560"; do
  echo "=== $cfg"; env $cfg timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --train-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/bench_ab.txt
timeout 240 python -m pytest tests/test_golden_gpu.py -m gpu -x -q -k "sample_one_video" > $O/pytest_golden.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_golden.txt; tail -n 2 $O/pytest_golden.txt
