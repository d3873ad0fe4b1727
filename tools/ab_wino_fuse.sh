# A/B of the in-launch split-K reduction on one box: headline bench under (Winograd fused, KSW fused) = (1,1) (1,0) (0,0), alternating
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r05_r}; mkdir -p $O
for rep in 1 2; do
for v in "1 1" "1 0" "0 0"; do
  set -- $v
  LFDM_WINO_FUSE_REDUCE=$1 LFDM_KSW_FUSE_REDUCE=$2 timeout 300 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-gpu-eager-baseline --train-steps 0 --lfae-train-steps 0 > $O/bench_fuse$1$2.json 2> $O/bench_fuse$1$2.err
  python -c "import json; b=json.load(open('$O/bench_fuse$1$2.json')); print('wino=$1 ksw=$2', b['value'], b['ms_per_step'])"
done
done
