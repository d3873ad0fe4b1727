R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/p4; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p4/kt -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --train-steps 0 > $R/gpurun_out/p4/bench_prof.json 2> $R/gpurun_out/p4/kt.err
python $R/tools/rocpd_stats.py $(find $R/gpurun_out/p4/kt -name '*.db' | head -1) > $R/gpurun_out/p4/kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/p4/kt
head -24 $R/gpurun_out/p4/kernel_stats.txt
