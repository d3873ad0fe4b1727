"""Empty stand-in: the reference imports torchvision at module import time only
(video_flow_diffusion.py:8, :940 default argument)."""
from . import transforms, models  # noqa: F401
