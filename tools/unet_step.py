#!/usr/bin/env python
"""N eager UNet forwards at the C2 shape (for rocprofv3 --pmc passes: few dispatches, same kernels as bench.py).
Usage: unet_step.py [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
torch.manual_seed(1234)
model, _, _ = synth.build_flow_diffusion("cuda:0", img_size=32, num_frames=40, sampling_timesteps=100, timesteps=1000)
img, cond = synth.inputs(1, 128, seed=7)
cond = cond.cuda()
x = torch.randn(1, 259, 40, 32, 32, device="cuda")
x[:, 3:] = x[:, 3:, :1]
tt = torch.full((1,), 500, device="cuda")
with torch.no_grad():
    for _ in range(iters):
        model.unet.forward(x, tt, cond=cond)
torch.cuda.synchronize()
