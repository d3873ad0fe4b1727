"""Same import path as the reference's DM/datasets_mug.py (`from datasets_mug import MUG` in the training scripts)."""
from cvpr23_lfdm_amd.datasets import MUG, MUG_test  # noqa: F401
