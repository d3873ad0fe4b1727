#!/usr/bin/env python
"""LFAE final conv (64 -> 3, 7x7, sigmoid) at 320 frames of 128x128: 4x4x1-MFMA kernel vs the 32-column KSW tile."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvpr23_lfdm_amd import ops  # noqa: E402

n, h, cin = 320, 128, 64
x = torch.randn(n * h * h, cin, device="cuda")
wt = torch.randn(3, cin, 7, 7, device="cuda") * 0.02
bias = torch.randn(3, device="cuda")
wp, bp = ops.pack_smalln_weight(wt, bias)
w4 = ops.pack_conv_weight(torch.cat((wt, wt.new_zeros(1, cin, 7, 7)), 0))
b4 = torch.cat((bias, bias.new_zeros(1)))
out = torch.empty(n * h * h, 4, device="cuda")
out2 = torch.empty(n * h * h, 4, device="cuda")


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


ms_a = timeit(lambda: ops.conv2d_smalln_cl(x, wp, bp, 3, 7, n, h, h, act=ops.ACT_SIGMOID, out=out))
ms_b = timeit(lambda: ops.conv2d_cl(x, w4, 4, 7, 7, n, h, h, bias=b4, act=ops.ACT_SIGMOID, out=out2))
print("small-N 4x4x1: %.2f ms   KSW 32-column tile: %.2f ms   max diff %.2e" % (ms_a, ms_b, float((out[:, :3] - out2[:, :3]).abs().max())))
