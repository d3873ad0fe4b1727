#!/usr/bin/env python
"""Timing of the other BASELINE.json sampling configurations on ONE GPU (not the headline metric; bench.py is):
  configs[2]  MHAD 128x128, 40-frame DDPM-1000 sample, batch=16
  configs[4]  NATOPS 256x256 (64x64 latent), 40-frame DDIM-50, batch=4 per GPU (32 over 8 GPUs), upconv/reflect/learned null
Prints one JSON line per configuration (synthetic weights / inputs)."""
import contextlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402

which = sys.argv[1:] or ["c3", "c5"]
CONFIGS = {
    "c3": dict(name="configs[2] MHAD 128x128 DDPM-1000 batch=16", latent=32, image=128, batch=16, steps=1000, timesteps=1000, kw={}),
    "c5": dict(name="configs[4] NATOPS 256x256 DDIM-50 batch=4/GPU", latent=64, image=256, batch=4, steps=50, timesteps=1000,
               kw=dict(learn_null_cond=True, use_deconv=False, padding_mode="reflect")),
}
for key in which:
    c = CONFIGS[key]
    torch.manual_seed(1234)
    with contextlib.redirect_stdout(sys.stderr):
        m, _, _ = synth.build_flow_diffusion("cuda:0", img_size=c["latent"], num_frames=40, sampling_timesteps=c["steps"],
                                             timesteps=c["timesteps"], **c["kw"])
    img, cond = synth.inputs(c["batch"], c["image"], seed=7)
    m.set_sample_input(sample_img=img.cuda(), sample_text=cond.cuda())
    reps = 1 if key == "c3" else 2
    m.sample_one_video(cond_scale=1.0)          # warm-up (captures the step graph)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        m.sample_one_video(cond_scale=1.0)
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / reps
    out = m.sample_out_vid
    assert bool(torch.isfinite(out).all())
    print(json.dumps({"config": c["name"], "batch": c["batch"], "seconds_per_batch": round(sec, 3),
                      "videos_per_s": round(c["batch"] / sec, 3), "ms_per_unet_step": round(1e3 * sec / c["steps"], 2),
                      "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))
    del m
    torch.cuda.empty_cache()
