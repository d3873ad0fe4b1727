# round 3: F(4x4) Winograd in the LFAE decode: the training / generator parity tests with it active, then the B = 8 training step with and without
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03n}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_train_step.py tests/test_full_size_gpu.py tests/test_golden_gpu.py tests/test_end_to_end.py -m gpu -x -q -k "not c3 and not c5" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -n 4 $O/pytest.txt
for v in 1 0; do echo "LFDM_WINO4=$v"; LFDM_WINO4=$v timeout 200 python tools/train_step.py 6 8 2>&1 | grep -v amdgpu.ids | tail -n 1 | cut -c1-300; done | tee $O/train_ab.txt
