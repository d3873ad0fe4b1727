"""Same import path as the reference's DM/modules/video_flow_diffusion_model_multiGPU.py (functional forward / sample)."""
from cvpr23_lfdm_amd.flow_diffusion import FlowDiffusionFunctional as FlowDiffusion  # noqa: F401
