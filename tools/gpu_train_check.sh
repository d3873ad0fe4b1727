# training-side check: parity tests that touch the training step / LFAE predictors / optimizer / two ranks, then the step time and
# the longest launches of one step
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-tr}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_train_step.py tests/test_unet_train.py tests/test_lfae_predictors.py tests/test_optim.py tests/test_dp_two_ranks.py tests/test_autograd.py tests/test_full_size_gpu.py tests/test_svd2x2.py tests/test_ops_parity.py tests/test_end_to_end.py tests/test_golden_gpu.py -m gpu -x -q -k "not c3 and not c5 and (winograd or golden or affine or pooled or generator or train or lfae or svd or region or pixelwise or bg or optim or dp or autograd or unet or c4 or sample_one_video)" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -n 4 $O/pytest.txt
timeout 300 python tools/train_step.py 6 8 2>&1 | grep -v amdgpu.ids | tail -n 1 > $O/train.txt; cat $O/train.txt
bash tools/prof_train.sh $TAG > $O/train_prof.txt 2>&1; head -n 40 $O/train_top_launches.txt
