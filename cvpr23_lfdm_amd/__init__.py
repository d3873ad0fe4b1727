"""cvpr23_lfdm_amd - MI355X-native LFDM sampling + DM-training path.

Public surface mirrors the reference's classes:
  FlowDiffusion (DM/modules/video_flow_diffusion_model.py), Unet3D / GaussianDiffusion
  (DM/modules/video_flow_diffusion.py), Generator (LFAE/modules/generator.py).
"""
from .diffusion import GaussianDiffusion  # noqa: F401
from .flow_diffusion import FlowDiffusion, FlowDiffusionFunctional  # noqa: F401
from .generator import Generator  # noqa: F401
from .unet import Unet3D  # noqa: F401
