#!/usr/bin/env python
"""gn_apply (GroupNorm finalize + scale/shift + SiLU) at the UNet level shapes, B=1 T=40: us per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvpr23_lfdm_amd import ops  # noqa: E402

t = 40
for s, c in ((32, 64), (16, 128), (8, 256), (4, 512)):
    rows = t * s * s
    x = torch.randn(rows, c, device="cuda")
    gamma, beta = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    ss = torch.randn(1, 2 * c, device="cuda") * 0.1
    nchunk = max(rows // 160, 1)
    partial = torch.rand(nchunk, 16, device="cuda")
    ws = torch.empty(1 << 20, device="cuda")
    out = torch.empty_like(x)
    fn = lambda: ops.groupnorm_apply_cl(x, 1, gamma, beta, partial, nchunk, scale_shift=ss, out=out, ws=ws)
    for _ in range(5):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    print("F4=%s res %2d C=%3d: %.2f us (%.0f GB/s)" % (os.environ.get("LFDM_GN_F4", "4"), s, c, us, 2 * rows * c * 4 / us / 1e3))
