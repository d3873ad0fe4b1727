#!/usr/bin/env python
"""Where does a Winograd convolution launch spend its time?  Uses a probe copy of the library built with
-DLFDM_WINO_TIMING (conv_wino.hip stores cycle stamps of every workgroup: entry, index set-up done, first patch
transformed, K loop done, epilogue done + a 100 MHz wall-clock span for calibration) and prints the mean phase lengths.
Build the probe here (no GPU needed):  python tools/probe_wino_phases.py --build ;  run on the GPU box without arguments."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "cvpr23_lfdm_amd", "csrc")
out = os.path.join(ROOT, "cvpr23_lfdm_amd", "build", "liblfdm_probe_wino.so")
if "--build" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm",
                           "-amdgpu-mfma-vgpr-form", "-DLFDM_WINO_TIMING", "-o", out] + sorted(glob.glob(os.path.join(csrc, "*.hip"))))
    sys.exit(0)
os.environ["LFDM_HIP_LIB"] = os.environ.get("LFDM_PROBE_LIB", out)
import torch  # noqa: E402
from cvpr23_lfdm_amd import ops  # noqa: E402

for cin, cout, s in ((64, 64, 32), (128, 64, 32), (128, 128, 16), (256, 256, 8), (512, 256, 8), (512, 512, 4), (1024, 512, 4), (256, 256, 32)):
    frames = 40
    m = frames * s * s
    x = torch.randn(m, cin, device="cuda")
    raw = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    w, ww = ops.pack_conv_weight(raw), ops.pack_wino_weight(raw)
    o = torch.empty(m, cout, device="cuda")
    pp, _ = ops.conv_params(x, w, cout, 3, 3, frames, s, s, out=o, weight_wino=ww)
    buf = torch.zeros(32768 + 16384 * 8, dtype=torch.int64, device="cuda")      # 64 K ticket words, then eight stamps per workgroup
    stamps = buf[32768:]
    pp.tile_counters, pp.tile_counters_len = buf.data_ptr(), 65536               # (the sampler's plan: split-K reduced inside the launch)
    rows, ks = ops.conv_plan(pp)
    if ops.conv_partial_floats(pp) > 0:                                           # (also the balanced ksplit = 1 launches)
        part = torch.empty(ops.conv_partial_floats(pp), device="cuda")
        pp.partial = part.data_ptr()
    for _ in range(3):
        ops.conv_launch(pp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv_launch(pp)
    e1.record()
    torch.cuda.synchronize()
    raw = stamps.cpu().view(-1, 8)
    raw = raw[raw[:, 0] > 0]
    st = raw.double()
    span_cyc = st[:, 4] - st[:, 0]
    tick = float((st[:, 5] * 0.01).sum() / span_cyc.sum())           # us per cycle (wall clock = 100 MHz)
    ph = (st[:, 1:5] - st[:, 0:4]).mean(dim=0) * tick
    nchunk = cin // 16 // ks
    print("3x3 %3d->%3d @%2d ksplit %d: %4d workgroups, %d chunks each | event %.1f us | setup %.2f  first patch %.2f  K-loop rest %.2f (%.2f/chunk)  "
          "epilogue %.2f | workgroup life %.2f us (max %.2f) | %.3f GHz" % (
              cin, cout, s, ks, st.shape[0], nchunk, e0.elapsed_time(e1) * 100, ph[0], ph[1], ph[2], float(ph[2]) / max(1, nchunk), ph[3],
              float(span_cyc.mean()) * tick, float(span_cyc.max()) * tick, 1e-3 / tick))
    if "--placement" in sys.argv:
        # where the dispatcher put the workgroups: HW_ID (cu_id bits 11:8, sh_id 12, se_id 15:13) | XCC_ID << 32, in linear workgroup-id order
        hw = raw[:, 6]
        cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (((hw >> 32) & 0xF) << 8)
        ids, counts = torch.unique(cu, return_counts=True)
        hist = torch.bincount(counts)
        print("      placement: %d CUs used; workgroups per CU -> number of CUs: %s" % (ids.numel(), {int(k): int(v) for k, v in enumerate(hist) if v}))
        print("      first 24 workgroups (linear id): (xcc, se, sh, cu) = %s" % [(int((h >> 32) & 0xF), int((h >> 13) & 7), int((h >> 12) & 1), int((h >> 8) & 0xF)) for h in hw[:24]])
        same = [int((cu == cu[i]).nonzero().flatten().tolist().__len__()) for i in range(4)]
        print("      workgroups sharing a CU with workgroup 0: ids %s" % (cu == cu[0]).nonzero().flatten().tolist())
        print("      workgroups sharing a CU with workgroup 1: ids %s" % (cu == cu[1]).nonzero().flatten().tolist())
        life = (raw[:, 4] - raw[:, 0]).double() * tick
        for n in sorted(set(counts.tolist())):
            sel = torch.isin(cu, ids[counts == n])
            print("      CUs with %d workgroups: mean workgroup life %.2f us, K loop %.2f us" % (n, float(life[sel].mean()), float(((raw[:, 3] - raw[:, 2]).double() * tick)[sel].mean())))
