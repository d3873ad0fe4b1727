"""Same import path as the reference's DM/modules/video_flow_diffusion_model.py."""
from cvpr23_lfdm_amd.flow_diffusion import FlowDiffusion  # noqa: F401
