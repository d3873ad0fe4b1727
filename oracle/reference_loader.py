"""TEST INFRASTRUCTURE ONLY - never imported by the product package.

Imports the *unmodified* reference implementation from /root/reference on a CPU-only
box (recipe: SURVEY.md Appendix C).  Used only by oracle/make_golden.py and by tests that
pin oracle/lfdm_oracle.py against the real reference.  /root/reference does not exist on
the GPU box, so nothing that runs there may call `load_reference()`.
"""
import os
import sys

REFERENCE_ROOT = os.environ.get("LFDM_REFERENCE_ROOT", "/root/reference")
_SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_shims")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "DM", "modules", "video_flow_diffusion.py"))


class _Ref:
    pass


_cached = None


def load_reference():
    """Returns a namespace with the reference modules (`vfd`, `vfdm`, `generator`, ...)."""
    global _cached
    if _cached is not None:
        return _cached
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    import torch
    from torch import nn

    # The repo root also has drop-in packages called DM/ and LFAE/ (the product's mirror of
    # the reference import paths).  The real reference must win inside this process.
    for name in list(sys.modules):
        if name == "DM" or name.startswith("DM.") or name == "LFAE" or name.startswith("LFAE."):
            del sys.modules[name]
    repo_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != repo_root]
    sys.path.insert(0, REFERENCE_ROOT)
    sys.path.insert(0, _SHIMS)
    sys.path.append(repo_root)          # the product package stays importable, behind the reference

    if not torch.cuda.is_available():
        # the reference hard-codes .cuda() (video_flow_diffusion.py:440,560,
        # video_flow_diffusion_model.py:41,50,110-112,...)
        nn.Module.cuda = lambda self, *a, **k: self
        torch.Tensor.cuda = lambda self, *a, **k: self

    import importlib
    ref = _Ref()
    ref.vfd = importlib.import_module("DM.modules.video_flow_diffusion")
    ref.vfdm = importlib.import_module("DM.modules.video_flow_diffusion_model")
    ref.generator = importlib.import_module("LFAE.modules.generator")
    ref.util = importlib.import_module("LFAE.modules.util")
    ref.config_mug = os.path.join(REFERENCE_ROOT, "config", "mug128.yaml")
    ref.config_natops = os.path.join(REFERENCE_ROOT, "config", "natops128.yaml")
    _cached = ref
    return ref
