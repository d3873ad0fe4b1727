"""Same import path as the reference's LFAE/modules/generator.py."""
from cvpr23_lfdm_amd.generator import Generator  # noqa: F401
