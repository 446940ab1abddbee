"""SiLK image extractor, native on MI355X.

Drop-in for SiLKModel (reference core/modules/image_extractors/silk_extractor.py:78-257).  The
reference wires a vendored SiLK `Flow` graph; only its arithmetic matters here: ParametricVGG
without pooling, four stages of two Conv-ReLU-BN blocks (silk/backbones/superpoint/vgg.py:221-290),
detector head 128->128->1 (magicpoint.py:53-101) and descriptor head 128->128->128
(superpoint.py:22-64).  The module tree below reproduces the reference's state_dict keys,
including the unused `SILK_BACKBONE.*` copy and `model.descriptor_scale_factor`.
The reference loads `silk/pvgg-4.ckpt` in its constructor (:167-174); here weights arrive through
`load_state_dict` (keys already stripped of `_mods.model.` exactly as the reference strips them).
"""
import torch
from torch import nn

from .._base import NativeExtractor
from ..net.vgg import vgg_block


class _PVGG(nn.Module):
    def __init__(self, in_channels=1, channels=(64, 64, 128, 128)):
        super().__init__()
        chans = (in_channels,) + tuple(channels)
        self.layers = nn.ModuleList([nn.Sequential(vgg_block(chans[i - 1], chans[i], 3, True), vgg_block(chans[i], chans[i], 3, True))
                                     for i in range(1, len(chans))])


class _DetHead(nn.Module):
    def __init__(self, cin, lat, cout):
        super().__init__()
        self._detH1 = vgg_block(cin, lat, 3, True)
        self._detH2 = nn.Sequential(nn.Conv2d(lat, cout, 1), nn.BatchNorm2d(cout))


class _DescHead(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self._desH1 = vgg_block(cin, cout, 3, True)
        self._desH2 = nn.Sequential(nn.Conv2d(cout, cout, 1), nn.BatchNorm2d(cout))


class _Mods(nn.Module):
    def __init__(self):
        super().__init__()
        self._mods = nn.ModuleDict({"logits": _DetHead(128, 128, 1), "raw_descriptors": _DescHead(128, 128)})


class _Shared(nn.Module):
    def __init__(self):
        super().__init__()
        self._backbone = _PVGG()
        self._heads = _Mods()


class _SiLKVGG(nn.Module):
    def __init__(self, scale):
        super().__init__()
        self.backbone = _Shared()
        self.descriptor_scale_factor = nn.parameter.Parameter(torch.tensor(float(scale)), requires_grad=False)


class SiLKModel(NativeExtractor):
    kind = "silk"
    cell_size = 1
    uses_batchnorm = True
    dilate_mask = False

    def __init__(self, device, padding, nms_radius=4, detection_top_k=2048, detection_threshold=0.0005, remove_borders=4,
                 ordering="yx", descriptor_scale_factor=1.0, learnable_descriptor_scale_factor=False):
        super().__init__()
        if padding not in (0, 1):
            raise AssertionError(padding)  # silk_extractor.py:147
        self.device = device
        self.padding = padding
        self._init_common(nms_radius, detection_top_k, detection_threshold, remove_borders, ordering, descriptor_scale_factor,
                          learnable_descriptor_scale_factor)
        self.SILK_SCALE_FACTOR = 1.41
        self.SILK_BACKBONE = _PVGG()  # unused copy the reference also registers (silk_extractor.py:107-111)
        self.model = _SiLKVGG(self.SILK_SCALE_FACTOR)

    def _stacks(self):
        bb = []
        for stage in self.model.backbone._backbone.layers:
            bb += [(stage[0], False), (stage[1], False)]
        heads = self.model.backbone._heads._mods
        return bb, [heads["logits"]._detH1, heads["logits"]._detH2], [heads["raw_descriptors"]._desH1, heads["raw_descriptors"]._desH2]

    def forward(self, image, *args, **kwargs):
        # the reference's SiLKModel.forward ignores any mask argument (:177)
        return super().forward(image, None)

    def extract_batched(self, x, score_mask=None, **kw):
        return super().extract_batched(x, None, **kw)

    def _prepare_input(self, image):
        out = image.clone()  # `image = image / 255.0` leaves the caller's tensor untouched (:178)
        from .... import _native as N
        return N.div_inplace(out, 255.0)
