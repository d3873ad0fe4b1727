# the C5 fixture at the configuration's own 50 DDIM steps (tests/golden/sample_ddim50_c5_256.npz, minted by the unmodified reference)
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03w}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
LFDM_PARITY_LOG=$O/parity.jsonl timeout 300 python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -k "c5" > $O/pytest_c5.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest_c5.txt
timeout 30 python tools/parity_margins.py $O/parity.jsonl $O/parity_margins.json | tail -n 2; grep -h "fraction_of_bar\|\"what\"\|\"test\"" $O/parity_margins.json | head -40
