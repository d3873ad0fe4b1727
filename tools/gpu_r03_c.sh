# round 3, third GPU call: the one-launch low-resolution attention blocks (attn_lowres.hip) - parity, A/B end to end, step sequence
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03c}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_parity.py -m gpu -x -q -k "lowres or attention or pointwise" > $O/pytest_ops.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_ops.txt; tail -n 3 $O/pytest_ops.txt
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_end_to_end.py -m gpu -x -q > $O/pytest_golden.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_golden.txt; tail -n 3 $O/pytest_golden.txt
for cfg in "LFDM_LOWRES_ATTN=0" "LFDM_LOWRES_ATTN=1"; do
  echo "=== $cfg"; env $cfg timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --train-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/bench_ab.txt
bash tools/prof_sequence.sh $TAG > $O/prof.txt 2>&1; tail -n 1 $O/step_sequence.txt
