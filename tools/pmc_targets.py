#!/usr/bin/env python
"""Launch sequences for the rocprofv3 --pmc passes behind bench.py's `roofline.traffic` fields (tools/prof_traffic.sh):
  wino  : the Winograd convolution launches (+ their split-K reduces) of one sampler step, N times
  warp  : the five warp launches of one 40-frame decode, N times
A 2-step sampler run first collects the step's launches (few dispatches: counter collection serialises every kernel); a
`zero_u32`-free marker is not needed - tools/pmc_traffic.py takes the LAST N x launches dispatches of each kernel."""
import contextlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import synth  # noqa: E402
from cvpr23_lfdm_amd import ops  # noqa: E402

target, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.manual_seed(1234)
with contextlib.redirect_stdout(sys.stderr):
    model, _, _ = synth.build_flow_diffusion("cuda:0", img_size=32, num_frames=40, sampling_timesteps=2, timesteps=1000)
img, cond = synth.inputs(1, 128, seed=7)
img, cond = img.cuda(), cond.cuda()
model.set_sample_input(sample_img=img, sample_text=cond)
if target == "wino":
    convs = [p for p in bench.sampler_step_convs(model) if p.weight_wino and p.kh == 3]
    nred = 0
    for p in convs:
        nred += 1 if ops.conv_plan(p)[1] > 1 else 0
    torch.cuda.synchronize()
    for _ in range(n):
        for p in convs:
            ops.conv_launch(p)
    torch.cuda.synchronize()
    print("PMC_TARGET wino launches_per_iter=%d reduce_per_iter=%d iters=%d" % (len(convs), nred, n))
else:
    bench.warp_bench(model, img, iters=n)
    print("PMC_TARGET warp launches_per_iter=5 iters=%d (+1 warm-up iteration)" % n)
