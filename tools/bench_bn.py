#!/usr/bin/env python
"""BatchNorm(batch statistics)+ReLU kernels of LFAE stage-1 training (csrc/train_lfae.hip) per shape: device time of each of the four
kernels (torch.profiler device events) and the HBM-side bytes they move, beside two plain streaming passes over the same tensor
(lfdm_absmax_f32 = read once, a device-to-device copy = read + write once).  Shapes = the generator's BatchNorm layers at 32 pairs.
Usage: bench_bn.py [--reps 5]"""
import argparse
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvpr23_lfdm_amd import lfae_ops as L  # noqa: E402
from cvpr23_lfdm_amd.ops import _lib, _p, _stream  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--warm", action="store_true", help="no cache flush between passes; x is re-written by a copy right before the forward (as a producing convolution would)")
a = ap.parse_args()
dev = "cuda:0"
SHAPES = [(32 * 128 * 128, 64), (32 * 128 * 128, 128), (32 * 64 * 64, 256), (32 * 64 * 64, 128), (32 * 32 * 32, 256), (96 * 32 * 32, 64)]
lib = _lib()
for rows, c in SHAPES:
    x = torch.randn(rows, c, device=dev)
    dy = torch.randn(rows, c, device=dev)
    g, b = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    amax = torch.zeros(4, device=dev)
    other = torch.empty_like(x)
    flush = torch.empty(96 << 20, device=dev)          # 384 MB: pushes the tensors out of the 256 MB infinity cache between passes

    xsrc = x.clone()

    def between():
        if not a.warm:
            flush.zero_()

    def run():
        if a.warm:
            x.copy_(xsrc)
        between()
        y, stat = L.batchnorm_train_fwd(x, g, b, None, None, 0.1, 1e-5, True)
        between()
        L.batchnorm_train_bwd(x, dy, g, b, stat, True)
        between()
        lib.check(lib.lfdm_absmax_f32(_p(x), rows, c, c, _p(amax), _stream(lib)), "absmax")
        between()
        other.copy_(x)
    run()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(a.reps):
            run()
        torch.cuda.synchronize()
    acc = collections.OrderedDict()
    for ev in prof.events():
        if str(getattr(ev, "device_type", "")).endswith("CUDA"):
            acc.setdefault(ev.name, []).append(ev.device_time if hasattr(ev, "device_time") else ev.cuda_time)
    mb = rows * c * 4 / 1e6
    traffic = {"bn_reduce_kernel<0>": mb, "bn_apply_kernel<0>": 2 * mb, "bn_reduce_kernel<1>": 2 * mb, "bn_apply_kernel<1>": 3 * mb, "absmax": mb,
               "Memcpy": 2 * mb, "copy": 2 * mb}
    print("# rows %d x C %d  (%.1f MB per tensor; %s)" % (rows, c, mb, "no flush, x freshly written" if a.warm else
                                                    "every pass starts with the tensors flushed from the infinity cache"))
    for name, ts in acc.items():
        if "Fill" in name or "Memset" in name:
            continue
        key = next((k for k in traffic if k in name), None)
        us = sorted(ts)[len(ts) // 2]
        print("  %-44s %8.1f us %s" % (name.replace("(anonymous namespace)::", "")[:44], us,
                                      ("%7.2f TB/s (%5.0f MB)" % (traffic[key] / us, traffic[key])) if key else ""))
