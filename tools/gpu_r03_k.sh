R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03k}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_ops_parity.py -m gpu -x -q -k "conv2d or layernorm or pointwise or planar_in or groupnorm or deconv" > $O/pytest_ops.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_ops.txt; tail -n 2 $O/pytest_ops.txt
grep -q "rc=0" $O/pytest_ops.txt || exit 1
for v in 0 1; do LFDM_STEM_MFMA=$v timeout 100 python tools/bench_stem.py 2>&1 | grep -v amdgpu.ids; done | tee $O/bench_stem.txt
timeout 200 python tools/bench_pw.py 2>&1 | grep -v amdgpu.ids > $O/bench_pw.txt; tail -n 22 $O/bench_pw.txt | cut -c1-100
timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --train-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"
timeout 300 bash tools/prof_sequence.sh $TAG > $O/prof.txt 2>&1; tail -n 1 $O/step_sequence.txt
