#!/usr/bin/env python
"""What ONE more dependent launch costs INSIDE the replayed sampler step - measured, not read off a trace.

rocprofv3's kernel records of a graph-replayed, barrier-ordered chain carry begin(i+1) == end(i): the wait for the predecessor's cache
write-back, the dispatch and the wave ramp are INSIDE each record's duration, so `gap_us` of profiles/*_step_sequence.txt is identically 0
and says nothing about the boundary.  This probe adds N tiny dependent launches (one 64-thread workgroup adding 1.0 to one float) to the
captured step - after every convolution, after every convolution and every GroupNorm apply, or all at the end of the step - and divides the
change of the video time by the number of launches added: the in-situ price of a kernel boundary + an empty kernel, behind a kernel that
has just written its output (vs behind another empty kernel).  GPU only; writes nothing (tee it into profiles/)."""
import os
import sys
import time

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import synth  # noqa: E402
from cvpr23_lfdm_amd import ops  # noqa: E402


def video_ms(m, n=3):
    m.sample_one_video(cond_scale=1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        m.sample_one_video(cond_scale=1.0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


def main():
    dev = "cuda"
    tiny = torch.zeros(64, device=dev)
    counts = {"n": 0}
    real_conv, real_gn, real_step = ops.conv_launch, ops.groupnorm_apply_cl, ops.sampler_step
    mode = {"conv": False, "gn": False, "tail": 0}

    active = {"on": False}

    def bump():
        if active["on"]:                 # only inside GaussianDiffusion._sample: the LFAE encode / decode launches stay as they are
            tiny.add_(1.0)
            counts["n"] += 1

    def hook_sampler(m):
        inner = m.diffusion._sample

        def _sample(*a, **k):
            active["on"] = True
            try:
                return inner(*a, **k)
            finally:
                active["on"] = False
        m.diffusion._sample = _sample

    def conv_launch(p):
        real_conv(p)
        if mode["conv"]:
            bump()

    def gn_apply(*a, **k):
        r = real_gn(*a, **k)
        if mode["gn"]:
            bump()
        return r

    def sampler_step(*a, **k):
        r = real_step(*a, **k)
        for _ in range(mode["tail"]):
            bump()
        return r

    ops.conv_launch, ops.groupnorm_apply_cl, ops.sampler_step = conv_launch, gn_apply, sampler_step
    rows = []
    for label, cfg in (("baseline", dict(conv=False, gn=False, tail=0)),
                       ("+1 empty launch after every convolution", dict(conv=True, gn=False, tail=0)),
                       ("+1 after every convolution and every GroupNorm apply", dict(conv=True, gn=True, tail=0)),
                       ("+40 empty launches at the end of the step (empty after empty)", dict(conv=False, gn=False, tail=40)),
                       ("baseline again", dict(conv=False, gn=False, tail=0))):
        mode.update(cfg)
        m, _, _ = synth.build_flow_diffusion(dev, img_size=32, num_frames=40, sampling_timesteps=100)     # fresh model: its step graph is captured with the hooks in this mode
        img, cond = synth.inputs(1, 128)
        m.set_sample_input(sample_img=img.to(dev), sample_text=cond.to(dev))
        hook_sampler(m)
        counts["n"] = 0
        m.sample_one_video(cond_scale=1.0)            # captures (dry run + capture of the single step + the 10-step chunks)
        ms = video_ms(m)
        rows.append((label, ms))
        del m
        torch.cuda.empty_cache()
    base = 0.5 * (rows[0][1] + rows[-1][1])
    print("# C2 video (100 replayed steps), ms per video; added launches per step counted from the hook calls of one capture")
    for (label, ms), per_step in zip(rows, (0, None, None, 40, 0)):
        print("%-66s %8.2f ms" % (label, ms))
    # launches added per step: convolutions / GroupNorm applies of one step (counted on a fresh eager pass)
    mode.update(conv=True, gn=True, tail=0)
    os.environ["LFDM_NO_GRAPH"] = "1"
    m, _, _ = synth.build_flow_diffusion(dev, img_size=32, num_frames=40, sampling_timesteps=2)
    img, cond = synth.inputs(1, 128)
    m.set_sample_input(sample_img=img.to(dev), sample_text=cond.to(dev))
    hook_sampler(m)
    m.sample_one_video(cond_scale=1.0)
    counts["n"] = 0
    mode.update(conv=True, gn=False)
    m.sample_one_video(cond_scale=1.0)
    n_conv = counts["n"] / 2.0
    counts["n"] = 0
    mode.update(conv=True, gn=True)
    m.sample_one_video(cond_scale=1.0)
    n_both = counts["n"] / 2.0
    print("per step: %.0f convolution launches, %.0f with the GroupNorm applies" % (n_conv, n_both))
    for (label, ms), n in zip(rows[1:4], (n_conv, n_both, 40)):
        print("%-66s %+7.2f ms per video = %.2f us per added launch (%d per step)" % (label, ms - base, (ms - base) * 1e3 / (100 * n), n))


if __name__ == "__main__":
    main()
