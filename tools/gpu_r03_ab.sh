# focus_present_mask support: the reference-minted fixture through the HIP sampling executor, forward + every gradient of the training executor
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03ab}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
LFDM_PARITY_LOG=$O/parity.jsonl timeout 300 python -m pytest tests/test_golden_gpu.py tests/test_unet_train.py -m gpu -x -q -k "focus or unet_forward or train_grads" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.txt
timeout 30 python tools/parity_margins.py $O/parity.jsonl $O/parity_margins.json | tail -n 1
