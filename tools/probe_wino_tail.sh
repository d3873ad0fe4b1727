# What the in-launch split-K hand-off of conv_wino_kernel<FUSE> costs per video: two PROBE builds of the library (results wrong, timing only)
#   notail1: slab stores, then return           (no store drain, no ticket, no reduce, no epilogue of the reducer)
#   notail2: store drain + ticket, then return  (the last arriver does not read the slabs back)
# against the in-tree build, alternating on one box (tools/ab_lib.py).  Build the probes here (CPU container), run on the GPU box:
#   bash tools/probe_wino_tail.sh build ;  gpurun -- bash tools/probe_wino_tail.sh run
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
if [ "$1" = build ]; then
  for v in 1 2; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -DLFDM_PROBE_NOTAIL=$v -c cvpr23_lfdm_amd/csrc/conv_wino.hip -o scratch/conv_wino_notail$v.o -Wno-unused-result || exit 1
    objs=$(ls cvpr23_lfdm_amd/build/*.o | grep -v conv_wino.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/liblfdm_hip_notail$v.so $objs scratch/conv_wino_notail$v.o || exit 1
  done
  ls -la scratch/*.so
else
  for v in 1 2; do python tools/ab_lib.py r06_g_wino_tail$v --base scratch/liblfdm_hip_notail$v.so --rounds 2; done
fi
