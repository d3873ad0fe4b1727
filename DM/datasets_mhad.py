"""Same import path as the reference's DM/datasets_mhad.py."""
from cvpr23_lfdm_amd.datasets import MHAD, MHAD_test  # noqa: F401
