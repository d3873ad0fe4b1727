cd $GRAFT_REPO_ROOT
for K in 1 2 3 4 6; do echo "=== KSPLIT=$K"; KSPLIT=$K python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids | grep "3x3"; done
