# round 6, call A: the new GPU tests (graph robustness, fused-reduce stress) + the headline on today's box
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r06_a}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_lfae_train.py tests/test_ops_parity.py tests/test_lfae_predictors.py -m gpu -x -q -k "graphed or stress or splitk_reduced or defaults or lfae_train" > $O/pytest_new.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_new.txt; tail -n 5 $O/pytest_new.txt
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-eager-baseline --train-steps 0 --lfae-train-steps 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; b=json.load(open('$O/bench.json')); print(b['value'], b['ms_per_step'], json.dumps(b.get('roofline'))[:400])"
