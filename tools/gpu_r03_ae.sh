# conv (split-K) -> GroupNorm without the reduce launch (LFDM_GN_SPLITK=1): end-to-end A/B, parity of the op and of the goldens with it on
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03ae}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for v in 0 1 0 1; do echo -n "LFDM_GN_SPLITK=$v: "; LFDM_GN_SPLITK=$v timeout 100 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --train-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done | tee $O/ab.txt
LFDM_GN_SPLITK=1 timeout 200 python -m pytest tests/test_ops_parity.py tests/test_golden_gpu.py -m gpu -x -q -k "splitk_slabs or unet_forward or sample_one_video" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 2 $O/pytest.txt
