def disk(*a, **k):
    raise RuntimeError("skimage is not installed; stub used for import only")
