"""Stand-in for the few imageio calls of the LFDM scripts (see README.md in this directory)."""
from cvpr23_lfdm_amd.io_compat import imread, imsave, mimsave  # noqa: F401


class _V2:
    imread = staticmethod(imread)
    imwrite = staticmethod(imsave)
    mimsave = staticmethod(mimsave)


v2 = _V2()
imwrite = imsave
