# the other BASELINE.json sampling configurations + throughput mode on one GPU (not the headline)
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-oc}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 600 python tools/bench_configs.py c3 c5 2> $O/configs.err | tee $O/other_configs.json
timeout 300 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --train-steps 0 2> $O/b8.err | tee -a $O/other_configs.json | cut -c1-300
