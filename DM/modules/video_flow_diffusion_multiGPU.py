"""Same import path as the reference's DM/modules/video_flow_diffusion_multiGPU.py."""
from cvpr23_lfdm_amd.diffusion import GaussianDiffusion  # noqa: F401
from cvpr23_lfdm_amd.unet import Unet3D  # noqa: F401
