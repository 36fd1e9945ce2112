"""TEST INFRASTRUCTURE ONLY — stand-in for the one torchvision symbol the reference's LPIPS needs: `torchvision.models.vgg16`
(tokenizer/tokenizer_image/lpips.py:118-136 reads `.features[0:30]`).  torchvision is not installed here and its weights are not
downloadable, so the architecture is restated from torchvision's published configuration 'D'
(torchvision/models/vgg.py, cfgs['D'] = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']:
Conv2d(k=3, padding=1) + ReLU(inplace=True) per number, MaxPool2d(kernel_size=2, stride=2) per 'M', module indices 0..30), random-init.
The reference's own LPIPS / vgg16 wrapper / NetLinLayer / ScalingLayer classes then run UNMODIFIED on top of it
(oracle/make_golden.py gen_vqloss).  Nothing in the product package may import this file."""
import types

import torch.nn as nn

_CFG_D = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']


class _VGG(nn.Module):
    def __init__(self):
        super().__init__()
        layers, cin = [], 3
        for v in _CFG_D:
            if v == 'M':
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)


def vgg16(pretrained=False, **kwargs):
    """torchvision.models.vgg16(pretrained=...) — `pretrained` is accepted and ignored (no checkpoint offline: random init)."""
    return _VGG()


models = types.SimpleNamespace(vgg16=vgg16)
