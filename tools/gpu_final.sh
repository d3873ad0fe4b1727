# end-of-round evidence: the whole -m gpu suite, smoke(), the default bench, a profiled step (kernel table + dispatch sequence),
# the training kernel table.  Outputs under gpurun_out/$TAG (copy what should be judged into profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-final}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -n 4 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep "\[bench" $O/bench.err | tail -n 4
bash tools/prof_sequence.sh $TAG > $O/prof.txt 2>&1; tail -n 1 $O/step_sequence.txt
bash tools/prof_train.sh > $O/train_prof.txt 2>&1; cp gpurun_out/p3/train_kernel_stats.txt $O/train_kernel_stats.txt; grep -a value gpurun_out/p3/kt.err | tail -n 1 | cut -c1-200
