R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/p3; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p3/kt -o t -- python $R/tools/train_step.py 2 8 > /dev/null 2> $R/gpurun_out/p3/kt.err
python $R/tools/rocpd_stats.py $(find $R/gpurun_out/p3/kt -name '*.db' | head -1) --top 60 > $R/gpurun_out/p3/train_kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/p3/kt
grep -a "value" $R/gpurun_out/p3/kt.err | tail -1
head -45 $R/gpurun_out/p3/train_kernel_stats.txt
