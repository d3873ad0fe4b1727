# headline bench under different split-K slice lengths of the Winograd plan (with the in-launch reduction a slice costs less than it did
# when the constants were chosen: profiles/r02_c_ksplit_sweep.txt)
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r05_t}; mkdir -p $O
run() {
  env "$@" timeout 300 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-gpu-eager-baseline --train-steps 0 --lfae-train-steps 0 > $O/b.json 2> $O/b.err
  python -c "import json; b=json.load(open('$O/b.json')); print('$*', b['value'], b['ms_per_step'])"
}
run LFDM_WINO_SLICE_CHUNKS=5
run LFDM_WINO_SLICE_CHUNKS=4
run LFDM_WINO_SLICE_CHUNKS=3
run LFDM_WINO_SLICE_CHUNKS=6
run LFDM_WINO_SLICE_CHUNKS=4 LFDM_WINO_SPLIT_MIN_CHUNKS=8
run LFDM_WINO_SLICE_CHUNKS=5
