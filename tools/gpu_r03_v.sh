R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03v}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
W4_FROM=7 timeout 200 python tools/bench_wino4.py > $O/bench_wino4_b1.txt 2>&1; echo "bench rc=$?"; grep -v W4US $O/bench_wino4_b1.txt
