cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_r; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_parity.py tests/test_lfae_ops.py -m gpu -x -q -k "reduced_in_launch or fused_groupnorm or batchnorm or l1 or apply_optical" 2>&1 | tail -3
for v in 1 0 1 0; do
  LFDM_WINO_FUSE_REDUCE=$v timeout 300 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-gpu-eager-baseline --train-steps 0 --lfae-train-steps 0 > $O/bench_fuse$v.json 2> $O/bench_fuse$v.err
  python -c "import json; b=json.load(open('$O/bench_fuse$v.json')); print('fuse=$v', b['value'], b['ms_per_step'])"
done
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_end_to_end.py -m gpu -x -q 2>&1 | tail -3
