# end-of-round evidence: the whole -m gpu suite (with the achieved error of every comparison logged), smoke(), the default bench, bench.py --gpus 2
# self-launched on the one GPU, a profiled step (kernel table + dispatch sequence), whole-step counters, the training kernel table.  Every
# command under its own timeout.  Outputs under gpurun_out/$TAG (copy what should be judged into profiles/ as ${TAG}_*, and put the tag + the
# counter file into profiles/LATEST: bench.py quotes them beside its live figures).
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-final}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
# a CLEAN build of the library on this box first (every object recompiled, hipcc --offload-arch=gfx950): the evidence below is of a library
# built from the sources in this tree, not of the one that travelled with the snapshot
( time timeout 1500 python -m cvpr23_lfdm_amd._build hip --force ) > $O/clean_build.txt 2>&1; echo "clean build rc=$?" >> $O/clean_build.txt; tail -n 5 $O/clean_build.txt
python -c "from cvpr23_lfdm_amd import _build; print('build fingerprint', _build.source_fingerprint())" >> $O/clean_build.txt
rm -f $O/parity.jsonl
LFDM_C3_DRIFT_OUT=$O/c3_ddpm1000_drift.txt LFDM_PARITY_LOG=$O/parity.jsonl timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -n 4 $O/pytest_gpu.txt
timeout 60 python tools/parity_margins.py $O/parity.jsonl $O/parity_margins.json | tail -n 3
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
# (the CPU baseline leg is left to the driver's own bench run: ~2 minutes of host time that the evidence run does not need)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err      # the driver's own command line (minus the CPU baseline leg); echo "bench rc=$?"; grep "\[bench" $O/bench.err | tail -n 4
timeout 300 bash tools/prof_sequence.sh $TAG > $O/prof.txt 2>&1; tail -n 1 $O/step_sequence.txt
timeout 400 bash tools/prof_step_pmc.sh $TAG > $O/pmc.txt 2>&1; head -n 3 $O/step_pmc.txt
# the N = 2 code path of bench.py on ONE GPU: plain `python bench.py --gpus 2` re-executes itself under torch.distributed.run (both ranks on
# cuda:0, gloo because fewer GPUs than ranks): barrier / max-over-ranks protocol, rank-0-only extras, the training step with the gradient
# all-reduce across two ranks and its exposed-communication figure
timeout 400 python bench.py --gpus 2 --steps 3 --warmup 1 --train-steps 3 --lfae-train-steps 2 --no-cpu-baseline --no-roofline --no-gpu-eager-baseline > $O/bench_n2_one_gpu.json 2> $O/bench_n2_one_gpu.err; echo "bench n2 rc=$?"; tail -c 700 $O/bench_n2_one_gpu.json; echo
timeout 300 bash tools/prof_train.sh > $O/train_prof.txt 2>&1; cp gpurun_out/p3/train_kernel_stats.txt gpurun_out/p3/train_top_launches.txt $O/; grep -a value gpurun_out/p3/kt.err | tail -n 1 | cut -c1-200
# LFAE stage-1 training (lfae_train.py; also a leg of bench.py): launches per step / native share from torch.profiler, then the rocprofv3 kernel table
timeout 200 python tools/lfae_census.py --batch 32 --top 30 > $O/lfae_census.txt 2> $O/lfae_census.err; head -n 6 $O/lfae_census.txt
timeout 300 bash tools/prof_lfae.sh $TAG > $O/lfae_prof.txt 2>&1; tail -n 3 $O/lfae_prof.txt
