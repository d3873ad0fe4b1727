#!/usr/bin/env python
"""One eager (LFDM_NO_GRAPH=1) 40-frame sample with 3 sampler steps: the dispatch stream the whole-step counter passes of
tools/prof_step_pmc.sh collect (rocprofv3 --pmc serialises every kernel, so the run is kept short).  tools/pmc_video_report.py
cuts the LAST sampler step out of it (the dispatches between the last two sampler_update_kernel launches) and the warp launches
of the decode that follows."""
import contextlib
import os
import sys

os.environ["LFDM_NO_GRAPH"] = "1"
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402

torch.manual_seed(1234)
with contextlib.redirect_stdout(sys.stderr):
    model, _, _ = synth.build_flow_diffusion("cuda:0", img_size=32, num_frames=40, sampling_timesteps=3, timesteps=1000)
img, cond = synth.inputs(1, 128, seed=7)
model.set_sample_input(sample_img=img.cuda(), sample_text=cond.cuda())
model.sample_one_video(cond_scale=1.0)
torch.cuda.synchronize()
print("PMC_VIDEO done: 3 sampler steps + LFAE decode, eager")
