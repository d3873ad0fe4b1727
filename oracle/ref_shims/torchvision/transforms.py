class _Unavailable:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        raise RuntimeError("torchvision is not installed; stub used for import only")


ToTensor = _Unavailable
ToPILImage = _Unavailable
functional = None
