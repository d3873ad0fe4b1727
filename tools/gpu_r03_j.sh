R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03j}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for v in 0 1; do LFDM_STEM_MFMA=$v timeout 100 python tools/bench_stem.py 2>&1 | grep -v amdgpu.ids; done | tee $O/bench_stem.txt
timeout 200 python -m pytest tests/test_end_to_end.py -m gpu -x -q -k "multi_step" > $O/pytest_e2e.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_e2e.txt; tail -n 3 $O/pytest_e2e.txt
for cfg in "LFDM_GRAPH_STEPS=1" "LFDM_GRAPH_STEPS=10" "LFDM_GRAPH_STEPS=25" "LFDM_GRAPH_STEPS=100"; do
  echo "=== $cfg"; env $cfg timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --train-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/bench_ab.txt
