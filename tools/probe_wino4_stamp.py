#!/usr/bin/env python
"""Who waits for whom inside conv_wino4_kernel?  Needs a probe build of csrc/conv_wino4.hip with -DLFDM_W4_STAMP (the role loops then
accumulate s_memtime cycles per wave: consumers = [MFMA work | period-barrier wait], producers = [wait for the patch loads | B^T d B
transform + LDS stores | issue of the next 36 loads | barrier wait]) linked into a library of its own and selected with LFDM_HIP_LIB:
  cd cvpr23_lfdm_amd && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -DLFDM_W4_STAMP -c csrc/conv_wino4.hip \
     -o build/probe/conv_wino4_stamp.o && hipcc --offload-arch=gfx950 -shared -fPIC -o build/probe/w4_stamp.so $(ls build/*.o | grep -v conv_wino4.o) build/probe/conv_wino4_stamp.o
  LFDM_HIP_LIB=$PWD/build/probe/w4_stamp.so [LFDM_W4_PRIO=0|1] python tools/probe_wino4_stamp.py
The stamps travel through the otherwise unused lfdm_conv_params.gn_in_gamma pointer.  Shape: 3x3 256 -> 256 on 320 frames of 32x32 (the
LFAE bottleneck of a B = 8 training step).  Results: profiles/r04_w_wino4_stamps.txt.  GPU only."""
import os, sys, ctypes
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["LFDM_WINO4_MIN"] = "1"
from cvpr23_lfdm_amd import ops
n, cin, cout, h, w = 320, 256, 256, 32, 32
dev = "cuda"
g = torch.Generator().manual_seed(1)
x = torch.randn(n * h * w, cin, generator=g).to(dev)
wt = (torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5)).to(dev)
res = torch.randn(n * h * w, cout, generator=g).to(dev)
ww, w4 = ops.pack_wino_weight(wt), ops.pack_wino4_weight(wt)
out = torch.empty(n * h * w, cout, device=dev)
pp, _ = ops.conv_params(x, None, cout, 3, 3, n, h, w, residual=res, out=out, weight_wino=ww, weight_wino4=w4)
staged = os.environ.get("LFDM_W4_STAGED", "1") != "0"
stamps = torch.zeros(256 * 8 * 8 if staged else 2 * 256 * 8 * 2, dtype=torch.int64, device=dev)
pp.gn_in_gamma = stamps.data_ptr()
for _ in range(3):
    ops.conv_launch(pp)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.conv_launch(pp); e1.record(); torch.cuda.synchronize()
print("launch us", e0.elapsed_time(e1) * 1e3)
if staged:
    st = stamps.cpu().view(256, 8, 8).double().mean(0) / 16
    print("STAGED flavour, cycles per period (mean over 256 workgroups):")
    for wv in range(4):
        print("consumer wave %d: MFMA phase %6.0f | barrier 1 %6.0f | phase B + barrier 2 %6.0f" % (wv, st[wv, 0], st[wv, 2], st[wv, 4]))
    for wv in range(4, 8):
        print("producer wave %d: patches + transform %6.0f | raw store + next loads %6.0f | barrier 1 %6.0f | V store %6.0f | barrier 2 %6.0f" % (
            wv, st[wv, 0], st[wv, 1], st[wv, 2], st[wv, 3], st[wv, 4]))
    sys.exit(0)
st = stamps.cpu()[:4096].view(256, 8, 2).double(); s2 = stamps.cpu()[4096:].view(256, 8, 2).double()
print("per-wave cycle sums over the 16-period loop (mean over 256 workgroups):")
for wv in range(8):
    print("wave %d (%s): work %8.0f  barrier-wait %8.0f   per period: work %6.0f wait %6.0f" % (wv, "consumer" if wv < 4 else "producer", st[:, wv, 0].mean(), st[:, wv, 1].mean(), st[:, wv, 0].mean() / 16, st[:, wv, 1].mean() / 16))
for wv in range(4, 8):
    print("producer wave %d per period: load wait %6.0f  transform+store %6.0f  fetch issue %6.0f" % (wv, s2[:, wv, 0].mean() / 16, s2[:, wv, 1].mean() / 16, st[:, wv, 0].mean() / 16))
