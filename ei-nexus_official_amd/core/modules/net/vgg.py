"""Parameter containers with the reference's module tree (state_dict key compatibility only).

These torch modules are never *called*: the forward pass runs in HIP (csrc/conv.hip).  They exist
so that `load_state_dict` of a reference checkpoint works unchanged, e.g.
`backbone.l1.0.0.weight` (conv) and `backbone.l1.0.2.running_var` (BatchNorm).
Reference layout: core/modules/net/vgg.py:5-46 (block = Conv2d, ReLU, BatchNorm2d).
"""
from torch import nn


def vgg_block(in_channels, out_channels, kernel_size, use_batchnorm=True, non_linearity="relu", padding=1):
    if non_linearity != "relu":
        raise NotImplementedError
    mods = [nn.Conv2d(in_channels, out_channels, kernel_size, padding=padding), nn.ReLU(inplace=True)]
    if use_batchnorm:
        mods.append(nn.BatchNorm2d(out_channels))
    return nn.Sequential(*mods)


def block_spec(block, pool=False):
    """(conv, bn-or-None, relu) of a vgg_block / head tail -> arguments for _native.ConvLayer."""
    conv = block[0]
    bn = None
    relu = False
    for m in list(block)[1:]:
        if isinstance(m, nn.ReLU):
            relu = True
        elif isinstance(m, nn.BatchNorm2d):
            bn = m
    return conv, bn, relu, pool
