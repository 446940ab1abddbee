"""Native drop-in for the reference package `core.modules` (its __init__ exposes `build_model`,
core/modules/__init__.py:5-12): model name -> constructor, same NotImplementedError for unknown names."""
from .EIM import EIM
from .ImageImageMatcher import ImageImageMatcher

_MODELS = {"EIM": EIM, "ImageImageMatcher": ImageImageMatcher}


def build_model(config, device, logger):
    try:
        ctor = _MODELS[config.name]
    except KeyError:
        raise NotImplementedError(f"Unsupported model: {config.name}") from None
    return ctor(config, device, logger)
