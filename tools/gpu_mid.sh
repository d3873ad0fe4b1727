# mid-round check of the training paths: the GPU tests of the training operators / steps, the two training legs of bench.py, the LFAE census.
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-mid}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_host_and_abi.py tests/test_lfae_ops.py tests/test_autograd.py tests/test_train_ops.py tests/test_unet_train.py tests/test_train_step.py tests/test_lfae_train.py -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -n 4 $O/pytest_gpu.txt
timeout 600 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-gpu-eager-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; b=json.load(open('$O/bench.json')); print(b['value'], b['ms_per_step']); print(json.dumps(b.get('train'))[:600]); print(json.dumps(b.get('lfae_train'))[:900])"
timeout 200 python tools/lfae_census.py --batch 32 --top 30 > $O/lfae_census.txt 2> $O/lfae_census.err; head -n 6 $O/lfae_census.txt
