#!/usr/bin/env python
"""conv_wgrad_kernel per shape (the DM training step at B = 8 and the LFAE stage-1 step at 32 pairs), event-timed incl. the slab reduce:
direct-form TFLOP/s against the 157.3 fp32-MFMA peak.  Usage: bench_wgrad.py [--reps 5]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvpr23_lfdm_amd import train_ops  # noqa: E402

SHAPES = [  # (label, cin, cout, k, n_img, res)
    ("unet L0 64->64", 64, 64, 3, 320, 32), ("unet L0 128->64", 128, 64, 3, 320, 32), ("unet L1 128->128", 128, 128, 3, 320, 16),
    ("unet L2 256->256", 256, 256, 3, 320, 8), ("unet L3 512->512", 512, 512, 3, 320, 4), ("unet qkv 64->768 1x1", 64, 768, 1, 320, 32),
    ("unet down 64->64 k4s2", 64, 64, 4, 320, 32),
    ("lfae bottleneck 256->256", 256, 256, 3, 32, 32), ("lfae 128->64 @128", 128, 64, 3, 32, 128), ("lfae 64->128 @128", 64, 128, 3, 32, 128),
    ("lfae 512->512 @8", 512, 512, 3, 32, 8),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--one", action="store_true")
    a = ap.parse_args()
    dev = "cuda"
    print("# %-28s %10s %10s %8s" % ("shape", "us", "TFLOP/s", "frac"))
    for label, cin, cout, k, n, res in SHAPES:
        stride = 2 if k == 4 else 1
        pad = 1 if k in (3, 4) else 0
        hq = (res + 2 * pad - k) // stride + 1
        x = torch.randn(n * res * res, cin, device=dev)
        dy = torch.randn(n * hq * hq, cout, device=dev)
        out = torch.empty(cout, cin, k, k, device=dev) if k * k <= 16 else None
        db = torch.empty(cout, device=dev) if out is not None else None
        fn = lambda: train_ops.conv_wgrad(x, dy, n, res, res, hq, hq, k, k, stride=stride, pad=(pad, pad), out=out, ci_off=0, dbias=db)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps
        fl = 2.0 * n * hq * hq * cin * cout * k * k
        print("  %-28s %10.1f %10.1f %8.3f" % (label, us, fl / us / 1e6, fl / us / 1e6 / 157.3), flush=True)


if __name__ == "__main__":
    if os.environ.get("LFDM_WGRAD3") is None and "--one" not in sys.argv:
        import subprocess
        for knob in ("1", "0"):
            print("# LFDM_WGRAD3=%s (%s)" % (knob, "nine-tap tile kernel for 3x3 / stride 1" if knob == "1" else "per-tap kernel everywhere"), flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(os.environ, LFDM_WGRAD3=knob))
    else:
        main()
