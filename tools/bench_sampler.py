#!/usr/bin/env python
"""lfdm_sampler_step_f32 at the C2 latent (B=1, 3x40x32x32): us per step (graph replay)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvpr23_lfdm_amd import ops  # noqa: E402

for b, shape in ((1, (3, 40, 32, 32)), (16, (3, 40, 32, 32)), (4, (3, 40, 64, 64))):
    x, eps, noise = [torch.randn(b, *shape, device="cuda") for _ in range(3)]
    n = x[0].numel()
    table = torch.rand(4, 6, device="cuda")
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    ws = ops.sampler_ws(b, n, "cuda")
    fn = lambda: ops.sampler_step(x, eps, noise, table, step, ws=ws, advance=False)
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print("B=%2d n=%7d: %.1f us per sampler step" % (b, n, e0.elapsed_time(e1) * 1e3 / 100))
