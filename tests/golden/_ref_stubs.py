"""Import shims that let the *reference* package (``/root/reference``) import in
this container.  TEST INFRASTRUCTURE ONLY: used by ``gen_golden.py`` to capture
golden vectors.  Nothing here travels into the product path, and nothing here
performs arithmetic on the hot path except ``torchvision...resize`` which is
expressed as the exact ``F.interpolate`` call torchvision 0.17 makes for
float tensors (only used for the dense ``normalized_descriptors`` output) and
``kornia.color.rgb_to_grayscale`` (round 6), restated from kornia 0.7.1 -- the
reference's pinned version -- for float inputs (only reached by 3-channel images).

Packages the reference imports at module import time but which are absent
here (requirements.txt of the reference): omegaconf, hydra, kornia, cv2,
torchvision, pytorch_lightning, skimage, loguru, pynvml.
"""
import sys
import types

import torch
import torch.nn.functional as F
import yaml


class AttrDict(dict):
    """Minimal stand-in for omegaconf.DictConfig (attribute + item access)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(obj):
    if isinstance(obj, dict):
        return AttrDict({k: to_attr(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [to_attr(v) for v in obj]
    return obj


def _merge(a, b):
    out = AttrDict({k: to_attr(v) for k, v in dict(a).items()})
    for k, v in dict(b).items():
        if k in out and isinstance(out[k], dict) and isinstance(v, dict):
            out[k] = _merge(out[k], v)
        else:
            out[k] = to_attr(v)
    return out


class _OmegaConf:
    @staticmethod
    def merge(*cfgs):
        out = AttrDict()
        for c in cfgs:
            out = _merge(out, c)
        return out

    @staticmethod
    def create(d=None):
        return to_attr(d or {})

    @staticmethod
    def load(path):
        with open(path) as f:
            return to_attr(yaml.safe_load(f))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    if "omegaconf" in sys.modules and hasattr(sys.modules["omegaconf"], "_einx_stub"):
        return
    _mod("omegaconf", OmegaConf=_OmegaConf, DictConfig=AttrDict, ListConfig=list, _einx_stub=True)
    _mod("cv2")
    k = _mod("kornia")
    def rgb_to_grayscale(image, rgb_weights=None):
        # kornia 0.7.1 (the reference's pin, requirements.txt:56), kornia/color/gray.py, float inputs: weights
        # tensor([0.299, 0.587, 0.114]) of the image's dtype, `w_r * r + w_g * g + w_b * b` on the [..., 3, H, W] tensor
        if image.shape[-3] != 3:
            raise ValueError(f"Input size must have a shape of (*, 3, H, W). Got {image.shape}")
        w_r, w_g, w_b = torch.tensor([0.299, 0.587, 0.114], device=image.device, dtype=image.dtype).unbind()
        r, g, b = image[..., 0:1, :, :], image[..., 1:2, :, :], image[..., 2:3, :, :]
        return w_r * r + w_g * g + w_b * b

    kc = _mod("kornia.color", rgb_to_grayscale=rgb_to_grayscale)
    k.color = kc
    sk = _mod("skimage")
    sk.io = _mod("skimage.io")
    h = _mod("hydra")
    h.utils = _mod("hydra.utils")
    _mod("pynvml")
    class _Log:
        def __getattr__(self, name):
            return lambda *a, **k: 0

    lg = _mod("loguru", logger=_Log())
    lg._defaults = _mod("loguru._defaults", LOGURU_FORMAT="")

    class LightningModule(torch.nn.Module):
        pass

    _mod("pytorch_lightning", LightningModule=LightningModule)

    class InterpolationMode:
        BILINEAR = "bilinear"
        NEAREST = "nearest"

    def resize(img, size, interpolation=InterpolationMode.BILINEAR, max_size=None, antialias=None):
        # torchvision 0.17 tensor path for float input: F.interpolate(..., align_corners=False,
        # antialias=False when antialias is None)
        return F.interpolate(img, size=list(size), mode="bilinear", align_corners=False, antialias=False)

    tv = _mod("torchvision")
    tvt = _mod("torchvision.transforms")
    tvf = _mod("torchvision.transforms.functional", InterpolationMode=InterpolationMode, resize=resize)
    tv.transforms = tvt
    tvt.functional = tvf
