# One gpurun call = one table: tools/bench_conv.py under each setting of the convolution experiment knobs, so a round's
# A/B questions cost ~10 s of GPU each instead of one call each.  Usage on the GPU box:
#   bash tools/sweep_conv.sh            -> gpurun_out/sweep_conv.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
{
  for cfg in "LFDM_WINO=1" "LFDM_WINO=1 LFDM_WINO_STAGE=1" "LFDM_WINO=1 LFDM_WINO_BN=64" "LFDM_WINO=1 LFDM_WINO_BN=64 LFDM_WINO_STAGE=1" "LFDM_WINO=1 LFDM_WINO_WIDE=1" \
             "LFDM_WINO=0" "LFDM_WINO=0 LFDM_CONV_FORCE=igemm"; do
    echo "=== $cfg"
    env $cfg python $R/tools/bench_conv.py 2>&1 | grep -v amdgpu.ids
  done
} > $O/sweep_conv.txt
grep -E "^===|weighted|faster" $O/sweep_conv.txt
