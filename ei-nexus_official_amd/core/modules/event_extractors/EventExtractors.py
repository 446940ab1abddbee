"""Event-voxel extractors, native on MI355X.

Drop-in for the reference classes (same constructor arguments, state_dict keys and output dict):
  VGGExtractor    core/modules/event_extractors/EventExtractors.py:437-624  (SuperPoint-shaped, cell 8)
  VGGExtractorNP  core/modules/event_extractors/EventExtractors.py:238-434  (SiLK-shaped, cell 1)
"""
from .._base import NativeExtractor
from ..net.backbone import VGGBackBone
from ..net.descriptor_head import VGGDescriptorHead
from ..net.detector_head import VGGDetectorHead


class _VGGBase(NativeExtractor):
    dilate_mask = True  # events_mask is dilated 3x3 before masking the score (:544-550)
    _pooling = True

    def __init__(self, in_channels, feat_channels, descriptor_dim, nms_radius, detection_top_k, detection_threshold=0.0005,
                 remove_borders=4, ordering="yx", descriptor_scale_factor=1.0, learnable_descriptor_scale_factor=False,
                 use_batchnorm=True, padding=1):
        super().__init__()
        self.descriptor_dim = descriptor_dim
        self.padding = padding
        self.uses_batchnorm = bool(use_batchnorm)
        self._init_common(nms_radius, detection_top_k, detection_threshold, remove_borders, ordering, descriptor_scale_factor,
                          learnable_descriptor_scale_factor)
        self.backbone = VGGBackBone(in_channels=in_channels, feat_channels=feat_channels, use_batchnorm=use_batchnorm,
                                    use_max_pooling=self._pooling, padding=padding)
        self.cell_size = 8 if self._pooling else 1
        self.detector_head_dim = self.cell_size ** 2 + 1 if self._pooling else 1
        self.detector_head = VGGDetectorHead(in_channels=feat_channels, lat_channels=256, out_channels=self.detector_head_dim,
                                             use_batchnorm=use_batchnorm, padding=padding)
        self.descriptor_head = VGGDescriptorHead(in_channels=feat_channels, out_channels=descriptor_dim, use_batchnorm=use_batchnorm,
                                                 padding=padding)

    def _stacks(self):
        return (self.backbone.layer_blocks(), [self.detector_head._detH1, self.detector_head._detH2],
                [self.descriptor_head._desH1, self.descriptor_head._desH2])


class VGGExtractor(_VGGBase):
    kind = "vgg"
    _pooling = True


class VGGExtractorNP(_VGGBase):
    kind = "vgg_np"
    _pooling = False
