# round 3: B = 8 training step with the F(4x4) Winograd LFAE decode (512-thread version) against F(2x2)
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03q}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for v in 1 0; do echo "LFDM_WINO4=$v"; LFDM_WINO4=$v timeout 200 python tools/train_step.py 8 8 2>&1 | grep -v amdgpu.ids | tail -n 1 | cut -c1-200; done | tee $O/train_ab.txt
