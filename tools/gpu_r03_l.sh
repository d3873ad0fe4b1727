# round 3, exploratory + counter call: (1) the bf16 x3 split micro-benchmark (report-only), (2) whole-step counter
# passes + the warp launches' counters (VERDICT r2 #4b, #7).  Every command under its own timeout.
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03l}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 150 tools/ubench/bf16x3_gemm.bin > $O/bf16x3_gemm.txt 2>&1; echo "bf16x3 rc=$?"; cat $O/bf16x3_gemm.txt
timeout 200 python -c "import torch; print(torch.cuda.get_device_name(0))"
bash tools/prof_step_pmc.sh $TAG
