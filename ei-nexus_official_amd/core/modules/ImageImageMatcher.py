"""ImageImageMatcher: image extractor on both sides + matcher (reference
core/modules/ImageImageMatcher.py:13-85)."""
import torch
from torch import nn

from ..._native import on_input_device
from .Extractors import ImageKeypointsExtractor
from .Matchers import Matcher


class ImageImageMatcher(nn.Module):
    def __init__(self, config, device="cuda", logger=None):
        super().__init__()
        self.device = device
        self.config = config
        self.logger = logger
        self.image_extractor = ImageKeypointsExtractor(config, logger, device=device)
        self.matcher = Matcher(config, logger, device=device)
        if config.pretrain_stage1.model_path is not None:
            m = torch.load(config.pretrain_stage1.model_path, map_location=device)
            self.image_extractor.load_state_dict({k[16:]: v for k, v in m.items() if "image_extractor" in k})
        if config.pretrain_stage2.model_path is not None:
            m = torch.load(config.pretrain_stage2.model_path, map_location=device)
            self.matcher.load_state_dict({k[8:]: v for k, v in m.items() if "matcher" in k})

    @on_input_device
    def forward(self, image0, image1, mask=None):
        f0 = self.image_extractor(image0, mask=mask)
        f1 = self.image_extractor(image1)
        matches = self.matcher(f0, f1) if self.matcher.matcher is not None else None
        return f0, f1, matches

    def count_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)
