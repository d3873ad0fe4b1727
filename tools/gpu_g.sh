R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02_g; mkdir -p $O; cd $R
timeout 200 python bench.py --no-cpu-baseline --train-steps 0 > $O/bench_a.json 2> $O/bench_a.err; echo "bench_a rc=$?"; tail -n 6 $O/bench_a.err; head -c 2500 $O/bench_a.json; echo
python tools/probe_wino_phases.py 2>&1 | grep -v amdgpu.ids > $O/wino_phases.txt; cat $O/wino_phases.txt
python tools/ubench/launch_floor.py 2>&1 | grep -v amdgpu.ids > $O/launch_floor.txt; head -n 5 $O/launch_floor.txt
for H in 1 2 4 8; do echo "HPB=$H"; LFDM_TATTN_HPB=$H python tools/bench_attn.py 2>&1 | grep -A2 "fused LN + qkv + temporal"; done
timeout 400 python bench.py --steps 2 --train-steps 0 --no-roofline > $O/bench_b.json 2> $O/bench_b.err; echo "bench_b rc=$?"; tail -n 12 $O/bench_b.err; tail -c 1200 $O/bench_b.json
