"""Extractor wrappers (reference core/modules/Extractors.py:32-217): config dispatch, device
placement and the freeze -> no_grad + eval policy."""
import torch
from torch import nn

from .event_extractors.EventExtractors import VGGExtractor, VGGExtractorNP
from .image_extractors.silk_extractor import SiLKModel
from .image_extractors.superpoint_extractor import SuperPointv1


def _finish(wrapper, extractor, device, logger, label):
    wrapper.extractor = extractor
    extractor.to(device)
    if wrapper.freeze:
        for p in extractor.parameters():
            p.requires_grad = False
        extractor.eval()
    else:
        extractor.train()
    if logger is not None:
        n_all = sum(p.numel() for p in extractor.parameters())
        logger.log_info(f"{label} - type: {wrapper.config.type} - freeze: {wrapper.config.freeze} - all_params: {n_all}")


class EventKeypointsExtractor(nn.Module):
    def __init__(self, config, logger=None, device="cuda"):
        super().__init__()
        self.config = config.event_extractor
        self.extractor = None
        self.device = device
        self.freeze = self.config.freeze
        self.extractor_type = self.config.type
        self.representation = None
        if self.extractor_type == "vgg":
            c = self.config.vgg
            ext = VGGExtractor(in_channels=c.in_channels, feat_channels=c.feat_channels, descriptor_dim=c.descriptor_dim,
                               nms_radius=c.nms_radius, detection_threshold=c.detection_threshold, detection_top_k=c.detection_top_k,
                               remove_borders=c.remove_borders, ordering=c.ordering, descriptor_scale_factor=c.descriptor_scale_factor,
                               learnable_descriptor_scale_factor=c.learnable_descriptor_scale_factor, use_batchnorm=c.use_batchnorm)
        elif self.extractor_type == "vgg_np":
            c = self.config.vgg_np
            ext = VGGExtractorNP(in_channels=c.in_channels, feat_channels=c.feat_channels, descriptor_dim=c.descriptor_dim,
                                 nms_radius=c.nms_radius, detection_threshold=c.detection_threshold, detection_top_k=c.detection_top_k,
                                 remove_borders=c.remove_borders, ordering=c.ordering, descriptor_scale_factor=c.descriptor_scale_factor,
                                 learnable_descriptor_scale_factor=c.learnable_descriptor_scale_factor, use_batchnorm=c.use_batchnorm,
                                 padding=c.padding)
        else:
            raise ValueError(f"Unsupported extractor type: {self.extractor_type}")
        _finish(self, ext, device, logger, "EventKeypointsExtractor")

    def extract_batched(self, events, score_mask=None, **kw):
        with torch.no_grad():
            return self.extractor.extract_batched(events, score_mask, **kw)

    def forward(self, events, score_mask=None):
        # inference only: gradients never flow through the native kernels
        with torch.no_grad():
            return self.extractor(events, score_mask)


class ImageKeypointsExtractor(nn.Module):
    def __init__(self, config, logger, device="cuda"):
        super().__init__()
        self.config = config.image_extractor
        self.extractor = None
        self.freeze = self.config.freeze
        self.device = device
        self.feature_type = self.config.type
        if self.feature_type == "superpointv1":
            c = self.config.superpointv1
            ext = SuperPointv1(descriptor_dim=c.descriptor_dim, nms_radius=c.nms_radius, detection_threshold=c.detection_threshold,
                               detection_top_k=c.detection_top_k, ordering=c.ordering, remove_borders=c.remove_borders,
                               descriptor_scale_factor=c.descriptor_scale_factor,
                               learnable_descriptor_scale_factor=c.learnable_descriptor_scale_factor)
        elif self.feature_type == "silk":
            ext = SiLKModel(device=self.device, **dict(self.config.silk))
        else:
            raise ValueError(f"Unsupported feature type: {self.feature_type}")
        _finish(self, ext, device, logger, "ImageKeypointsExtractor")

    def extract_batched(self, image, mask=None, **kw):
        assert image.dim() == 4, f"Expected 4D tensor, got {image.dim()}D tensor instead."
        with torch.no_grad():
            return self.extractor.extract_batched(image, mask, **kw)

    def forward(self, image, mask=None):
        assert image.dim() == 4, f"Expected 4D tensor, got {image.dim()}D tensor instead."
        with torch.no_grad():
            return self.extractor(image, mask)
