#!/usr/bin/env python
"""gn_apply (GroupNorm finalize + scale/shift + SiLU) at the shapes of one B = 1, T = 40 sampler step, with the partial-chunk counts
the producing convolutions really emit: us per call inside a replayed graph.  LFDM_GN_F4 / LFDM_GN_BLOCK are read once by the
library, so a sweep is one process per setting (tools/sweep_gn.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvpr23_lfdm_amd import ops  # noqa: E402

t = 40
tag = "F4=%s BLOCK=%s" % (os.environ.get("LFDM_GN_F4", "-"), os.environ.get("LFDM_GN_BLOCK", "-"))
# (resolution, channels, groups, partial chunks): Winograd epilogue = one per 128 rows, split-K reduce = one per 16 rows
for s, c, groups, nchunk in ((32, 64, 8, 320), (32, 128, 16, 320), (16, 128, 8, 80), (16, 64, 8, 640), (8, 256, 8, 20), (8, 256, 8, 160),
                             (4, 512, 8, 40), (4, 256, 8, 40)):
    rows = t * s * s
    x = torch.randn(rows, c, device="cuda")
    gamma, beta = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    ss = torch.randn(1, 2 * c, device="cuda") * 0.1
    partial = torch.rand(nchunk, 2 * groups, device="cuda")
    ws = torch.empty(1 << 20, device="cuda")
    out = torch.empty_like(x)
    fn = lambda: ops.groupnorm_apply_cl(x, 1, gamma, beta, partial, nchunk, scale_shift=ss, out=out, ws=ws, groups=groups)
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
    print("%-18s res %2d C=%3d G=%2d chunks %3d: %6.2f us (%.0f GB/s)" % (tag, s, c, groups, nchunk, us, 2 * rows * c * 4 / us / 1e3))
