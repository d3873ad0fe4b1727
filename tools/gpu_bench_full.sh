R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-bf}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_lfae_predictors.py -m gpu -x -q > $O/fullsize.txt 2>&1; tail -n 6 $O/fullsize.txt
bash tools/prof_traffic.sh > $O/traffic.log 2>&1; cp gpurun_out/pt/traffic.json $O/traffic.json 2>/dev/null; mkdir -p profiles; cp gpurun_out/pt/traffic.json profiles/r02_traffic.json 2>/dev/null; tail -n 3 $O/traffic.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 3500 $O/bench.json; tail -n 5 $O/bench.err
