#!/usr/bin/env python
"""Which torch (ATen) kernels does one DM training step launch besides the library's own, and from where?

Runs FlowDiffusion.optimize_parameters on a tiny shape (CPU: the kernels run in the x86 emulator, the torch-side op sequence
is the same as on the GPU) under a TorchDispatchMode and prints the ops grouped by the innermost cvpr23_lfdm_amd source line.
Usage: count_torch_ops.py [--device cpu|cuda] [--batch 1] [--frames 2] [--hw 32] [--top 60]"""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# views / metadata: no kernel behind them
FREE = {"view", "_unsafe_view", "reshape", "permute", "transpose", "t", "slice", "select", "expand", "unsqueeze", "squeeze",
        "detach", "alias", "as_strided", "split", "split_with_sizes", "unbind", "chunk", "narrow", "empty", "empty_like",
        "empty_strided", "new_empty", "new_empty_strided", "_local_scalar_dense", "unfold", "view_as", "lift_fresh",
        "is_same_size", "sym_size", "stride", "size", "numel", "dim", "storage_offset", "_reshape_alias", "set_", "resize_",
        "record_stream", "is_pinned", "_to_copy_noop"}


class Counter(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by_site = collections.Counter()
        self.by_op = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        if name not in FREE:
            site = "(autograd engine / torch internals)"
            for fr in reversed(traceback.extract_stack(limit=40)):
                if "cvpr23_lfdm_amd" in fr.filename and "count_torch_ops" not in fr.filename:
                    site = "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
                    break
            self.by_site[(site, name)] += 1
            self.by_op[name] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--hw", type=int, default=32)
    ap.add_argument("--top", type=int, default=60)
    a = ap.parse_args()
    import synth
    from cvpr23_lfdm_amd import FlowDiffusion
    dev = a.device
    if dev == "cpu":                        # same sources compiled for x86 against tests/emu (test infrastructure)
        from cvpr23_lfdm_amd import _build, _native
        _native._set_library_for_tests(_native.NativeLibrary(_build.build_emu(), "emu"))
    m = FlowDiffusion(img_size=a.hw // 4, num_frames=a.frames, sampling_timesteps=5, null_cond_prob=0.1, is_train=True, lr=1e-4,
                      config_pth=synth.CONFIG, pretrained_pth="")
    m.unet.load_state_dict(synth.unet_state())
    m.generator.load_state_dict(synth.generator_state())
    m.region_predictor.load_state_dict(synth.region_state())
    m.bg_predictor.load_state_dict(synth.bg_state())
    for net in (m.generator, m.region_predictor, m.bg_predictor):
        net.eval()
        m.set_requires_grad(net, False)
    m.to(dev)
    ref_img, real_vid, cond, _, _ = synth.train_inputs(a.batch, a.frames, a.hw)
    m.set_train_input(ref_img=ref_img.to(dev), real_vid=real_vid.to(dev), ref_text=cond.to(dev))
    m.optimize_parameters()                 # warm-up: packs, arenas
    c = Counter()
    with c:
        m.optimize_parameters()
    total = sum(c.by_op.values())
    print("# %d torch ops with a kernel behind them in one step" % total)
    for name, n in c.by_op.most_common(25):
        print("  %-32s %5d" % (name, n))
    print("# by call site")
    for (site, name), n in c.by_site.most_common(a.top):
        print("  %5d  %-28s %s" % (n, name, site))


if __name__ == "__main__":
    main()
