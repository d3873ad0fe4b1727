#!/usr/bin/env python
"""conv_planar_in (UNet stem over the 3 step-dependent channels, 7x7, 64 outputs, + per-video add term) at the sampler's shape,
standalone inside a replayed graph.  LFDM_STEM_MFMA=0/1 selects the VALU / matrix-pipe form (read once per process)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvpr23_lfdm_amd import ops  # noqa: E402

b, t, s = 1, 40, 32
x = torch.randn(b, 3, t, s, s, device="cuda")
w = ops.pack_planar_in_weight(torch.randn(64, 3, 7, 7, device="cuda") * 0.1)
add = torch.randn(b * s * s, 64, device="cuda")
out = torch.empty(b * t * s * s, 64, device="cuda")
fn = lambda: ops.conv_planar_in_cl(x, b, 3, 3, t, s, s, w, 7, 7, 64, add_term=add, out=out)
for _ in range(3):
    fn()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20):
        fn()
g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    g.replay()
e1.record()
torch.cuda.synchronize()
print("LFDM_STEM_MFMA=%s: %.2f us per launch" % (os.environ.get("LFDM_STEM_MFMA", "1"), e0.elapsed_time(e1) * 1e3 / 200))
