"""Same import path as the reference's DM/datasets_natops.py."""
from cvpr23_lfdm_amd.datasets import NATOPS, NATOPS_test  # noqa: F401
