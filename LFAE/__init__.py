"""Drop-in import path: `from DM.modules.video_flow_diffusion_model import FlowDiffusion` resolves to the
MI355X-native implementation when this repository root precedes the reference on sys.path."""
