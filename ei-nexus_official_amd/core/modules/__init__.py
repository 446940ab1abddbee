"""Drop-in for the reference's `core.modules` package (core/modules/__init__.py:1-12)."""
from .EIM import EIM
from .ImageImageMatcher import ImageImageMatcher


def build_model(config, device, logger):
    if config.name == "EIM":
        return EIM(config, device, logger)
    if config.name == "ImageImageMatcher":
        return ImageImageMatcher(config, device, logger)
    raise NotImplementedError(f"Unsupported model: {config.name}")
