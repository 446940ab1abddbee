"""ei-nexus_official_amd: MI355X-native event<->image feature extraction + matching.

The package mirrors the reference's `core.modules` API (EIM, extractors, matchers and the
detector/descriptor helper functions) on top of hand-written HIP kernels (csrc/, C ABI in
include/einx.h).  Importing it loads libeinx_hip.so and fails loudly if the library is missing.

    pkg = importlib.import_module("ei-nexus_official_amd")      # the directory name has a hyphen
    model = pkg.EIM(pkg.default_config("SP_MNN"), device="cuda")

To use it under the reference's import paths put the package directory on sys.path:
`sys.path.insert(0, ".../ei-nexus_official_amd"); from core.modules import build_model`.
"""
import os as _os

# see bench.py / INTEGRATION.md: with RCCL initialised, 4 hardware queues are not enough for the two-stream
# extractor schedule (effective when the HIP runtime has not been initialised yet; set it in the environment of
# the launcher otherwise)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _lib  # noqa: E402

_lib.load()  # no fallback: raise now if the HIP extension is not built

from . import _native as native  # noqa: E402
from . import configs, shard, synth  # noqa: E402
from .configs import AttrDict, default_config  # noqa: E402
from .core.modules import EIM, ImageImageMatcher, build_model  # noqa: E402
from .core.modules.Extractors import EventKeypointsExtractor, ImageKeypointsExtractor  # noqa: E402
from .core.modules.Matchers import Matcher  # noqa: E402
from .core.modules.matchers.MNN import NearestNeighborMatcher  # noqa: E402
from .core.modules.matchers.lightglue import LightGlue  # noqa: E402
from .harness import DifferentTimeEvaluator, SameTimeEvaluator  # noqa: E402



def install_as_core():
    """Alias this package's `core` tree as the top-level `core` package so that code written
    against the reference (`from core.modules import build_model`, `from core.modules.utils.
    detector_util import ...`) imports the native implementation unchanged."""
    import importlib
    import pkgutil
    import sys

    from . import core as _core
    sys.modules["core"] = _core
    for info in pkgutil.walk_packages(_core.__path__, _core.__name__ + "."):
        mod = importlib.import_module(info.name)
        sys.modules["core" + info.name[len(_core.__name__):]] = mod
    return _core


__all__ = ["install_as_core", "SameTimeEvaluator", "DifferentTimeEvaluator", "EIM", "ImageImageMatcher", "build_model", "EventKeypointsExtractor", "ImageKeypointsExtractor", "Matcher",
           "NearestNeighborMatcher", "LightGlue", "default_config", "AttrDict", "native"]
