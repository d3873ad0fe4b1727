# kernel-trace of the default bench (sampling only) -> gpurun_out/pk/kernel_stats.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pk; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --train-steps 0 > $O/bench_prof.json 2> $O/kt.err
python $R/tools/rocpd_stats.py $(find $O/kt -name '*.db' | head -1) > $O/kernel_stats.txt 2>&1
rm -rf $O/kt
head -40 $O/kernel_stats.txt
