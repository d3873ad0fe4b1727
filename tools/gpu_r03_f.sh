R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03f}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 200 python tools/bench_attn_lowres.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_attn_lowres.txt
timeout 300 bash tools/prof_sequence.sh $TAG > $O/prof.txt 2>&1; tail -n 1 $O/step_sequence.txt
