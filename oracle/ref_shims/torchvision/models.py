def vgg19(*a, **k):
    raise RuntimeError("torchvision is not installed; stub used for import only")
