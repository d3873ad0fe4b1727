R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-prof}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 300 bash tools/prof_sequence.sh $TAG > $O/prof.txt 2>&1; tail -n 1 $O/step_sequence.txt
