"""VGGDescriptorHead parameter container (reference: core/modules/net/descriptor_head.py:7-43)."""
from torch import nn

from .vgg import vgg_block


class VGGDescriptorHead(nn.Module):
    def __init__(self, in_channels=128, out_channels=256, use_batchnorm=True, padding=1):
        super().__init__()
        self._desH1 = vgg_block(in_channels, out_channels, 3, use_batchnorm, padding=padding)
        tail = [nn.Conv2d(out_channels, out_channels, 1, padding=0)]
        if use_batchnorm:
            tail.append(nn.BatchNorm2d(out_channels))
        self._desH2 = nn.Sequential(*tail)

    def forward(self, *a, **k):
        raise RuntimeError("parameter container only; the forward pass is native")
