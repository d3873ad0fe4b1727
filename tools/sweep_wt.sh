cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "LFDM_WINO_WT=32" "LFDM_WINO_WT=16" "LFDM_WINO_WT=16 KSPLIT=1" "LFDM_WINO_WT=16 KSPLIT=2" "LFDM_WINO_WT=16 KSPLIT=3" "LFDM_WINO_WT=16 KSPLIT=4"; do echo "=== $cfg"; env $cfg python tools/bench_conv.py 2>&1 | grep "3x3"; done
