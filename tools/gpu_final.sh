# end-of-round evidence: the whole -m gpu suite, smoke(), the default bench, a profiled step (kernel table + dispatch sequence),
# the training kernel table.  Outputs under gpurun_out/$TAG (copy what should be judged into profiles/ as ${TAG}_*, and put the tag into profiles/LATEST:
# bench.py quotes that run's rocprofv3 average beside its live figure).
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-final}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -n 4 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep "\[bench" $O/bench.err | tail -n 4
# the N = 2 code path of bench.py on ONE GPU (both ranks on cuda:0, gloo instead of RCCL): barrier / max-over-ranks protocol, rank-0-only extras,
# the training step with the gradient all-reduce across two ranks
LFDM_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --train-steps 3 > $O/bench_n2_one_gpu.json 2> $O/bench_n2_one_gpu.err; echo "bench n2 rc=$?"; tail -c 600 $O/bench_n2_one_gpu.json; echo
bash tools/prof_sequence.sh $TAG > $O/prof.txt 2>&1; tail -n 1 $O/step_sequence.txt
bash tools/prof_train.sh > $O/train_prof.txt 2>&1; cp gpurun_out/p3/train_kernel_stats.txt gpurun_out/p3/train_top_launches.txt $O/; grep -a value gpurun_out/p3/kt.err | tail -n 1 | cut -c1-200
