#!/usr/bin/env python
"""Which torch (ATen) kernels does one LFAE stage-1 training step (lfae_train.LFAETrainer.step) launch besides the library's own, and from
where?  CPU: the kernels run in the x86 emulator on the tiny configuration of tests/test_lfae_train.py - the torch-side op sequence per
layer is the same as on the GPU (the mug128 configuration has more layers of the same kinds).  Prints the ops grouped by the innermost
cvpr23_lfdm_amd source line.   Usage: count_lfae_ops.py [--kind tiny|mug128] [--device cpu|cuda] [--top 80]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from count_torch_ops import Counter  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--kind", default="tiny")
    ap.add_argument("--top", type=int, default=80)
    a = ap.parse_args()
    import synth
    import test_lfae_train as T
    if a.device == "cpu":
        from cvpr23_lfdm_amd import _build, _native
        _native._set_library_for_tests(_native.NativeLibrary(_build.build_emu(), "emu"))
    trainer, (mp, tp, hw, b) = T._build(a.kind, a.device)
    src, drv, theta, tps = synth.lfae_train_inputs(b, hw, tp)
    x = {"source": src.to(a.device), "driving": drv.to(a.device)}
    trainer.step(x, transform_noise=(theta, tps))
    c = Counter()
    with c:
        trainer.step(x, transform_noise=(theta, tps))
    total = sum(c.by_op.values())
    print("# %d torch ops with a kernel behind them in one step (%s)" % (total, a.kind))
    for name, n in c.by_op.most_common(40):
        print("  %-32s %5d" % (name, n))
    print("# by call site")
    for (site, name), n in c.by_site.most_common(a.top):
        print("  %5d  %-28s %s" % (n, name, site))


if __name__ == "__main__":
    main()
