# rocprofv3 kernel trace of 3 training steps (B = 8): per-kernel table + the longest launches of the last step
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-p3}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o t -- python $R/tools/train_step.py 2 8 > /dev/null 2> $O/kt.err
DB=$(find $O/kt -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB --top 60 > $O/train_kernel_stats.txt 2>&1
python $R/tools/rocpd_top.py $DB --top 70 > $O/train_top_launches.txt 2>&1
rm -rf $O/kt
grep -a "value" $O/kt.err | tail -1
head -30 $O/train_kernel_stats.txt
