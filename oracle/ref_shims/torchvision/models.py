"""torchvision is not installed here and there is no network for its ImageNet weights.  `vgg19(pretrained=True).features` - all that
LFAE/modules/model.py:26 takes - is rebuilt as torchvision defines it (Conv2d 3x3 pad 1 / ReLU / MaxPool2d 2, configuration "E") with the
repo's deterministic synthetic weights (cvpr23_lfdm_amd.params.synthetic_vgg19_state(seed) - what tests/synth.vgg_state() returns), so that the reference's perceptual loss and the product's see the same network.  TEST INFRASTRUCTURE ONLY."""
import torch
from torch import nn

_CFG_E = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
VGG_SEED = 1919


class _Vgg(nn.Module):
    def __init__(self):
        super().__init__()
        layers, cin = [], 3
        for v in _CFG_E:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)


def vgg19(pretrained=False, **_):
    from cvpr23_lfdm_amd import params as P
    m = _Vgg()
    sd = P.synthetic_vgg19_state(VGG_SEED)
    with torch.no_grad():
        for k, v in sd.items():
            parts = k.split(".")
            if len(parts) == 3:
                getattr(m.features[int(parts[1])], parts[2]).copy_(v)
    return m
