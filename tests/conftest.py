import os
import sys

import pytest

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO_ROOT not in sys.path:
    sys.path.insert(0, REPO_ROOT)
sys.path.insert(0, os.path.join(REPO_ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "emu: runs the HIP kernel sources under the x86 fiber emulator")


def _emu_enabled():
    return os.environ.get("LFDM_EMU", "1") != "0"


@pytest.fixture(params=[pytest.param("hip", marks=pytest.mark.gpu), pytest.param("emu", marks=pytest.mark.emu)])
def backend(request):
    """Selects which build of the kernels the ops layer drives.
    hip: liblfdm_hip.so on cuda:0 (the product).  emu: the same sources compiled for x86 against
    tests/emu (test infrastructure: checks index logic on a GPU-less box)."""
    import torch
    from cvpr23_lfdm_amd import _native
    if request.param == "hip":
        if not torch.cuda.is_available():
            pytest.skip("no GPU")
        _native._set_library_for_tests(None)
        yield "cuda"
    else:
        if not _emu_enabled():
            pytest.skip("LFDM_EMU=0")
        from cvpr23_lfdm_amd import _build
        path = _build.build_emu()
        _native._set_library_for_tests(_native.NativeLibrary(path, "emu"))
        yield "cpu"
        _native._set_library_for_tests(None)
