"""Stand-in for the few cv2 names of the LFDM scripts (see README.md in this directory)."""
import numpy as np

from cvpr23_lfdm_amd.io_compat import INTER_AREA, INTER_CUBIC, INTER_LINEAR, INTER_NEAREST, _resize_hw  # noqa: F401

BORDER_CONSTANT = 0


def resize(img, dsize, interpolation=INTER_LINEAR):
    """cv2.resize(img, (width, height), interpolation=...)"""
    return _resize_hw(img, dsize[1], dsize[0], interpolation)


def copyMakeBorder(img, top, bottom, left, right, border_type=BORDER_CONSTANT, value=0):
    pad = [(top, bottom), (left, right)] + [(0, 0)] * (np.asarray(img).ndim - 2)
    return np.pad(np.asarray(img), pad, mode="constant", constant_values=0)
