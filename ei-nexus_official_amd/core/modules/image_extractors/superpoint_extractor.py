"""SuperPoint v1 image extractor, native on MI355X.

Drop-in for SuperPointv1 (reference core/modules/image_extractors/superpoint_extractor.py:271-480):
same constructor, `conv1a..convDb` state_dict keys, output dict and the in-place `image /= 255`
side effect (:372).  The reference downloads pretrained weights in its constructor (:316-317);
this build has no network access by design -- load them with `load_state_dict`.
"""
from torch import nn

from .... import _native as N
from .._base import NativeExtractor


class SuperPointv1(NativeExtractor):
    kind = "superpointv1"
    cell_size = 8
    uses_batchnorm = False
    dilate_mask = False  # score_mask is used as given (:381-382, :411-412)
    input_div = 255.0    # `image /= 255.0` on the caller's tensor (:372)

    def __init__(self, descriptor_dim=256, nms_radius=4, detection_top_k=2048, detection_threshold=0.0005, remove_borders=4,
                 ordering="yx", descriptor_scale_factor=1.0, learnable_descriptor_scale_factor=False):
        super().__init__()
        self.descriptor_dim = descriptor_dim
        c1, c2, c3, c4, c5 = 64, 64, 128, 128, 256
        self.relu = nn.ReLU(inplace=True)
        self.pool = nn.MaxPool2d(kernel_size=2, stride=2)
        self.conv1a = nn.Conv2d(1, c1, 3, 1, 1)
        self.conv1b = nn.Conv2d(c1, c1, 3, 1, 1)
        self.conv2a = nn.Conv2d(c1, c2, 3, 1, 1)
        self.conv2b = nn.Conv2d(c2, c2, 3, 1, 1)
        self.conv3a = nn.Conv2d(c2, c3, 3, 1, 1)
        self.conv3b = nn.Conv2d(c3, c3, 3, 1, 1)
        self.conv4a = nn.Conv2d(c3, c4, 3, 1, 1)
        self.conv4b = nn.Conv2d(c4, c4, 3, 1, 1)
        self.convPa = nn.Conv2d(c4, c5, 3, 1, 1)
        self.convPb = nn.Conv2d(c5, 65, 1, 1, 0)
        self.convDa = nn.Conv2d(c4, c5, 3, 1, 1)
        self.convDb = nn.Conv2d(c5, descriptor_dim, 1, 1, 0)
        self._init_common(nms_radius, detection_top_k, detection_threshold, remove_borders, ordering, descriptor_scale_factor,
                          learnable_descriptor_scale_factor)

    def _spec(self, spec):
        conv, relu = spec
        return conv, None, relu

    def _layer(self, spec, pool=False):
        conv, relu = spec
        return N.ConvLayer(conv.weight, conv.bias, None, relu=relu, pool=pool)

    def _stacks(self):
        bb = [((self.conv1a, True), False), ((self.conv1b, True), True), ((self.conv2a, True), False), ((self.conv2b, True), True),
              ((self.conv3a, True), False), ((self.conv3b, True), True), ((self.conv4a, True), False), ((self.conv4b, True), False)]
        return bb, [(self.convPa, True), (self.convPb, False)], [(self.convDa, True), (self.convDb, False)]

    def _prepare_input(self, image):
        if image.dim() != 4:
            raise AssertionError(f"Expected 4D tensor, got {image.dim()}D tensor instead.")
        if image.shape[1] != 1:
            raise NotImplementedError("einx SuperPointv1 takes single-channel images (the EI-Nexus pipelines feed grayscale)")
        if not image.is_contiguous():
            raise RuntimeError("einx: image must be contiguous (it is scaled in place like the reference does)")
        return image  # `image /= 255.0` happens in place inside the extractor call (input_div)
