# round 3: the asm-pipelined projections of attn_lowres.hip - parity, then end-to-end A/B.  Every command has its own short timeout
# (a faulting kernel under rocprofv3 once hung for the whole call).
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03e}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 240 python -m pytest tests/test_ops_parity.py -m gpu -x -q -k "lowres" > $O/pytest_ops.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_ops.txt; tail -n 3 $O/pytest_ops.txt
grep -q "rc=0" $O/pytest_ops.txt || exit 1
timeout 240 python -m pytest tests/test_golden_gpu.py -m gpu -x -q -k "unet_forward or ddim5 or ddpm8" > $O/pytest_golden.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_golden.txt; tail -n 3 $O/pytest_golden.txt
for cfg in "LFDM_LOWRES_ATTN=0" "LFDM_LOWRES_ATTN=1"; do
  echo "=== $cfg"; env $cfg timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --train-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/bench_ab.txt
