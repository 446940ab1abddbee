"""SuperPoint v1 image extractor, native on MI355X.

Drop-in for SuperPointv1 (reference core/modules/image_extractors/superpoint_extractor.py:271-480):
same constructor, `conv1a..convDb` state_dict keys, output dict and the in-place `image /= 255`
side effect (:372).  The reference downloads pretrained weights in its constructor (:316-317);
this build has no network access by design -- load them with `load_state_dict`.
"""
import torch
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

    def _network_input(self, image, prepared):
        """`image /= 255.0` on the CALLER's tensor, then `rgb_to_grayscale` for 3-channel images (:372-376).  Contiguous
        single-channel images (what the EI-Nexus pipelines feed) are scaled in place inside einx_extract; RGB and / or
        non-contiguous images go through einx_image_prepare: scaled in place through their strides -- the caller sees the scaled
        tensor afterwards, as with the reference -- and the network reads a contiguous gray copy (kornia 0.7.1's weights)."""
        if image.dim() != 4:
            raise AssertionError(f"Expected 4D tensor, got {image.dim()}D tensor instead.")
        C = image.shape[1]
        if C == 1 and image.is_contiguous():
            return image, (0.0 if prepared else self.input_div)
        if C not in (1, 3):  # the reference scales in place, then conv1a refuses the tensor
            if not prepared and image.is_contiguous() and image.dtype is torch.float32 and image.device.type == "cuda":
                N.div_inplace(image, self.input_div)
            raise RuntimeError(f"Given groups=1, weight of size [64, 1, 3, 3], expected input{list(image.shape)} to have 1 channels, "
                               f"but got {C} channels instead")
        return N.image_prepare(image, 1.0 if prepared else self.input_div), 0.0
