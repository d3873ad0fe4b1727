# Where does a chunk of conv_wino4_kernel spend its time?  Probe builds (-DLFDM_W4_PROBE=<mask>: pipeline stages left out at COMPILE time;
# built in the container BEFORE the call (removed afterwards): for a in ...; hipcc -DLFDM_W4_PROBE=$a -c csrc/conv_wino4.hip; link with the other objects into
# cvpr23_lfdm_amd/build/probe/w4_$a.so) timed on the LFAE bottleneck shape: 1 = no patch loads, 2 = no transform (raw patch to LDS),
# 4 = no filter-fragment loads after chunk 0, 8 = no MFMAs (one v_fma instead), 16 = producers idle.  Results are wrong by construction.
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-w4abl}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for a in 0 1 2 8 16; do
  echo -n "mask=$a: "; W4_SHAPES=2 LFDM_HIP_LIB=$R/cvpr23_lfdm_amd/build/probe/w4_$a.so timeout 100 python tools/bench_wino4.py 2>/dev/null | grep W4US | tr '\n' ' '; echo
done | tee $O/ablate.txt
