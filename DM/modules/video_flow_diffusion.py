"""Same import path as the reference's DM/modules/video_flow_diffusion.py."""
from cvpr23_lfdm_amd.diffusion import GaussianDiffusion, cosine_beta_schedule  # noqa: F401
from cvpr23_lfdm_amd.unet import BERT_MODEL_DIM, Unet3D  # noqa: F401
