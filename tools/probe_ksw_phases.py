#!/usr/bin/env python
"""Where does a KSW convolution launch spend its time?  Builds a probe copy of the library with -DLFDM_KSW_TIMING
(conv_ksw.hip stores s_memtime stamps of every workgroup at: entry, after the index set-up, after the pipeline
prologue, after the K loop, after the epilogue), runs a few UNet-sized convolutions and prints the mean phase lengths
in microseconds (100 MHz constant clock of s_memtime) together with the spread of workgroup start / end times."""
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "cvpr23_lfdm_amd", "csrc")
out = os.path.join(tempfile.gettempdir(), "liblfdm_probe_timing.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm",
                       "-amdgpu-mfma-vgpr-form", "-DLFDM_KSW_TIMING"] + sys.argv[1:] + ["-o", out] + sorted(glob.glob(os.path.join(csrc, "*.hip"))))
os.environ["LFDM_HIP_LIB"] = out
import torch  # noqa: E402
from cvpr23_lfdm_amd import ops  # noqa: E402

TICK_US = 1.0 / 2100.0  # __builtin_readcyclecounter counts shader cycles (~2.1 GHz under MFMA load): calibrated against the event-timed kernel
for cin, cout, k, s in ((64, 64, 3, 32), (128, 128, 3, 16), (256, 64, 1, 32), (256, 256, 3, 32)):
    frames = 40
    m = frames * s * s
    x = torch.randn(m, cin, device="cuda")
    w = ops.pack_conv_weight(torch.randn(cout, cin, k, k, device="cuda") * 0.05)
    o = torch.empty(m, cout, device="cuda")
    pp, _ = ops.conv_params(x, w, cout, k, k, frames, s, s, out=o)
    rows, ks = ops.conv_plan(pp)
    if rows != 160 or ks != 1:
        print("skip", cin, cout, k, s, rows, ks)
        continue
    stamps = torch.zeros(8192 * 5, dtype=torch.int64, device="cuda")
    pp.partial = stamps.data_ptr()
    for _ in range(3):
        ops.conv_launch(pp)
    torch.cuda.synchronize()
    nblk = ((m + 159) // 160) * ((w.shape[1] + (63 if cout > 32 else 31)) // 64 if False else 1)
    st = stamps.cpu().view(-1, 5)
    st = st[st[:, 0] > 0].double()
    st = st[-(((m + 159) // 160) * max(1, w.shape[1] // (64 if w.shape[1] > 32 and ((m + 159) // 160) * ((w.shape[1] + 63) // 64) >= 224 else 32))):]
    t0 = st[:, 0].min()
    ph = (st[:, 1:] - st[:, :-1]).mean(dim=0) * TICK_US
    print("%dx%d %3d->%3d @%2d: blocks %4d | setup %.2f  prologue %.2f  K-loop %.2f  epilogue %.2f us | first start..last start %.2f us, "
          "first start..last end %.2f us" % (k, k, cin, cout, s, st.shape[0], ph[0], ph[1], ph[2], ph[3],
                                              float(st[:, 0].max() - t0) * TICK_US, float(st[:, 4].max() - t0) * TICK_US))
