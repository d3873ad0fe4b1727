# round 3, second GPU call: the tuned pointwise planner + GroupNorm-apply sizes end to end; parity margins of every golden comparison
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03b}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
rm -f $O/parity.jsonl
LFDM_PARITY_LOG=$O/parity.jsonl timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -n 4 $O/pytest_gpu.txt
python tools/parity_margins.py $O/parity.jsonl $O/parity_margins.json
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --train-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"
bash tools/prof_sequence.sh $TAG > $O/prof.txt 2>&1; tail -n 1 $O/step_sequence.txt
