# rocprofv3 kernel trace of LFAE stage-1 training steps (tools/train_lfae.py, config mug128, 32 synthetic 128x128 pairs): per-kernel table
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-plfae}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o t -- python $R/tools/train_lfae.py --batch 32 --steps 3 --warmup 2 --bench > $O/lfae_prof.json 2> $O/kt.err
DB=$(find $O/kt -name '*.db' | head -1)
(echo "# rocprofv3 --kernel-trace of tools/train_lfae.py --batch 32 --steps 3 --warmup 2 (config mug128, 128x128 synthetic pairs): 5 steps incl. set-up launches"; python $R/tools/rocpd_stats.py $DB --top 70) > $O/lfae_train_kernel_stats.txt 2>&1
rm -rf $O/kt
tail -1 $O/lfae_prof.json; head -12 $O/lfae_train_kernel_stats.txt
