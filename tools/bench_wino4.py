#!/usr/bin/env python
"""Winograd F(2x2,3x3) (conv_wino.hip, the plan's own tile choice) against F(4x4,3x3) (conv_wino4.hip) on the BATCHED 3x3 shapes: the
frozen-LFAE decode of a B = 8 training step (320 frames) and UNet convolutions at B = 8 / 16.  Checks the F(4x4) result against the
F(2x2) one first (max |diff| / max |out|), then times back-to-back launches with events.  GPU only."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvpr23_lfdm_amd import ops  # noqa: E402

SHAPES = [
    # (name, n_img, cin, cout, h, w, upsample, residual, act)
    ("LFAE bottleneck 256->256 @32 x320", 320, 256, 256, 32, 32, False, True, 0),
    ("LFAE bottleneck 256->256 @32 x320 relu", 320, 256, 256, 32, 32, False, False, 1),
    ("UpBlock2d 256->128 @32->64 x320", 320, 256, 128, 32, 32, True, False, 1),
    ("UpBlock2d 128->64 @64->128 x320", 320, 128, 64, 64, 64, True, False, 1),
    ("UNet 64->64 @32 x320 (B=8)", 320, 64, 64, 32, 32, False, False, 0),
    ("UNet 128->128 @16 x320 (B=8)", 320, 128, 128, 16, 16, False, False, 0),
    ("UNet 256->256 @8 x640 (B=16)", 640, 256, 256, 8, 8, False, False, 0),
    ("LFAE bottleneck 256->256 @32 x40 (B=1)", 40, 256, 256, 32, 32, False, True, 0),
    ("UNet 64->64 @32 x40 (B=1)", 40, 64, 64, 32, 32, False, False, 0),
    ("UNet 128->64 @32 x40 (B=1)", 40, 128, 64, 32, 32, False, False, 0),
    ("UNet 128->128 @16 x40 (B=1)", 40, 128, 128, 16, 16, False, False, 0),
]
if os.environ.get("W4_FROM"):
    SHAPES = SHAPES[int(os.environ["W4_FROM"]):]


def timed(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def timed_graph(fn, n=20, reps=10):
    """small launches: inside a replayed hipGraph (what the sampler does), us per launch"""
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


def main():
    dev = "cuda"
    lib = ops._lib()
    os.environ["LFDM_WINO"] = "1"
    os.environ["LFDM_WINO4_MIN"] = "1"
    print("%-44s %8s | %9s %6s | %9s %6s | %5s | %9s" % ("shape", "GFLOP", "F(2x2) us", "TF/s", "F(4x4) us", "TF/s", "x", "rel diff"))
    shapes = SHAPES[:int(os.environ["W4_SHAPES"])] if os.environ.get("W4_SHAPES") else SHAPES
    for name, n, cin, cout, h, w, up, resid, act in shapes:
        g = torch.Generator().manual_seed(1)
        x = torch.randn(n * h * w, cin, generator=g).to(dev)
        wt = (torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5)).to(dev)
        b = torch.randn(cout, generator=g).to(dev)
        ho, wo = (2 * h, 2 * w) if up else (h, w)
        res = torch.randn(n * ho * wo, cout, generator=g).to(dev) if resid else None
        ww, w4 = ops.pack_wino_weight(wt), ops.pack_wino4_weight(wt)
        outs, us = [], []
        for f4 in (False, True):
            os.environ["LFDM_WINO4"] = "1" if f4 else "0"
            out = torch.empty(n * ho * wo, cout, device=dev)
            pp, _ = ops.conv_params(x, None, cout, 3, 3, n, h, w, bias=b, residual=res, act=act, upsample=up, out=out, weight_wino=ww,
                                    weight_wino4=w4)
            kind = lib.lfdm_conv2d_schedule(ctypes.byref(pp))
            assert kind == (4 if f4 else 2), kind
            _, ks = ops.conv_plan(pp)
            keep = None
            if ks > 1:
                keep = torch.empty(ops.conv_partial_floats(pp), device=dev)
                pp.partial = keep.data_ptr()
            ops.conv_launch(pp)
            torch.cuda.synchronize()
            outs.append(out.clone())
            us.append(timed_graph(lambda: ops.conv_launch(pp)) if n < 320 else timed(lambda: ops.conv_launch(pp), 10))
        gf = 2.0 * n * ho * wo * cout * cin * 9 / 1e9
        diff = float((outs[0] - outs[1]).abs().max() / outs[0].abs().max())
        print("%-44s %8.1f | %9.1f %6.1f | %9.1f %6.1f | %5.2f | %9.2e" % (name, gf, us[0], gf / us[0] * 1e3, us[1], gf / us[1] * 1e3,
                                                                         us[0] / us[1], diff))
        print("W4US %.1f" % us[1])
        if diff > 5e-5:
            print("  !! F(4x4) and F(2x2) disagree")


if __name__ == "__main__":
    main()
