"""Same import path as the reference's top-level misc.py (`from misc import Logger, grid2fig, conf2fig, resize`)."""
from cvpr23_lfdm_amd.io_compat import (Logger, conf2fig, flow2fig, get_grid, grid2fig, resample, resize)  # noqa: F401
