#!/usr/bin/env python
"""Which call sites of one DM training step (B = 8) issue device-to-device memcpys and small ATen kernels?  torch.profiler with stacks;
prints per CPU op (the innermost cvpr23_lfdm_amd frame) the count of 'Memcpy DtoD' / copy kernels launched under it."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import synth  # noqa: E402
from cvpr23_lfdm_amd import FlowDiffusion  # noqa: E402

dev = "cuda:0"
torch.manual_seed(4321)
m = FlowDiffusion(img_size=32, num_frames=40, sampling_timesteps=1000, null_cond_prob=0.1, is_train=True, lr=1e-4, config_pth=synth.CONFIG, pretrained_pth="")
m.unet.load_state_dict(synth.unet_state())
m.generator.load_state_dict(synth.generator_state())
m.region_predictor.load_state_dict(synth.region_state())
m.bg_predictor.load_state_dict(synth.bg_state())
for net in (m.generator, m.region_predictor, m.bg_predictor):
    net.eval()
    m.set_requires_grad(net, False)
m.to(dev)
ref_img, real_vid, cond, _, _ = synth.train_inputs(8, 40, 128, seed=100)
m.set_train_input(ref_img=ref_img.to(dev), real_vid=real_vid.to(dev), ref_text=cond.to(dev))
for _ in range(2):
    m.optimize_parameters()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    m.optimize_parameters()
    torch.cuda.synchronize()
by_op = collections.Counter()
for ev in prof.events():
    if str(getattr(ev, "device_type", "")).endswith("CUDA") and ("Memcpy" in ev.name or "copyBuffer" in ev.name):
        by_op["(device) " + ev.name] += 1
# CPU-side ops that launched a memcpy: aten::copy_ events with their python stack
sites = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_foreach_copy_", "aten::cat") and ev.stack:
        site = next((s for s in ev.stack if "cvpr23_lfdm_amd" in s), ev.stack[0] if ev.stack else "?")
        sites[(ev.name, site.split("/")[-1][:110])] += 1
print(dict(by_op))
for (name, site), n in sites.most_common(40):
    print("%5d  %-22s %s" % (n, name, site))
