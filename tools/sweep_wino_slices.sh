# headline bench under different split-K slice lengths of the Winograd plan (with the in-launch reduction a slice costs less than it did
# when the constants were chosen: profiles/r02_c_ksplit_sweep.txt; round 6: again after the slabs became 16-byte write-through accesses)
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r05_t}; mkdir -p $O
run() {
  env "$@" timeout 300 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-gpu-eager-baseline --train-steps 0 --lfae-train-steps 0 --blocks 1 > $O/b.json 2> $O/b.err
  python -c "import json; b=json.load(open('$O/b.json')); print('%-70s %.4f videos/s %.2f ms' % ('$*', b['value'], b['ms_per_step']))" | tee -a $O/sweep.txt
}
rm -f $O/sweep.txt
run LFDM_NOOP=default
run LFDM_WINO_SLICE_CHUNKS=5
run LFDM_WINO_SLICE_CHUNKS=4
run LFDM_WINO_SLICE_CHUNKS=3
run LFDM_WINO_SLICE_CHUNKS=2
run LFDM_WINO_SLICE_CHUNKS=6
run LFDM_WINO_SLICE_CHUNKS=3 LFDM_WINO_SPLIT_MIN_CHUNKS=4
run LFDM_WINO_SLICE_CHUNKS=2 LFDM_WINO_SPLIT_MIN_CHUNKS=4
run LFDM_WINO_SPLIT_MIN_CHUNKS=4
run LFDM_WINO_SPLIT_MIN_CHUNKS=16
run LFDM_NOOP=default_again
