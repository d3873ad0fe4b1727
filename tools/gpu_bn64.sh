# A/B: 64-column Winograd workgroups on the batched (training) shapes
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-bn64}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for v in off 3072 1536 768; do
  if [ $v = off ]; then unset LFDM_WINO_BN64_MIN; else export LFDM_WINO_BN64_MIN=$v; fi
  echo "== LFDM_WINO_BN64_MIN=$v" >> $O/train.txt
  timeout 300 python tools/train_step.py 6 8 2>&1 | grep -v amdgpu.ids | tail -n 1 >> $O/train.txt
done
cat $O/train.txt
