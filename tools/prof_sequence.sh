# kernel trace of a short sampling run -> per-kernel table + the launch-by-launch sequence of one step
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-seq}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --train-steps 0 > $O/bench_prof.json 2> $O/kt.err
DB=$(find $O/kt -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB > $O/kernel_stats.txt 2>&1
python $R/tools/rocpd_sequence.py $DB > $O/step_sequence.txt 2>&1
rm -rf $O/kt
head -30 $O/kernel_stats.txt; tail -3 $O/step_sequence.txt
