"""VGGDetectorHead parameter container (reference: core/modules/net/detector_head.py:6-48)."""
from torch import nn

from .vgg import vgg_block


class VGGDetectorHead(nn.Module):
    def __init__(self, in_channels=128, lat_channels=256, out_channels=1, use_batchnorm=True, padding=1, detach=False):
        super().__init__()
        self._detH1 = vgg_block(in_channels, lat_channels, 3, use_batchnorm, padding=padding)
        tail = [nn.Conv2d(lat_channels, out_channels, 1, padding=0)]
        if use_batchnorm:
            tail.append(nn.BatchNorm2d(out_channels))
        self._detH2 = nn.Sequential(*tail)

    def forward(self, *a, **k):
        raise RuntimeError("parameter container only; the forward pass is native")
