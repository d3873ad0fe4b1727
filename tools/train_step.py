#!/usr/bin/env python
"""N DM training steps at B videos/GPU (for rocprofv3). Usage: train_step.py [steps] [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.cuda.set_device(0)
print(bench.train_bench("cuda:0", 0, 1, steps, 1, batch, lazy_extra=os.environ.get("LFDM_TRAIN_LAZY", "0") == "1"), file=sys.stderr)
