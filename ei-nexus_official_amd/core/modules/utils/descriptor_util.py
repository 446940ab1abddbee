"""Descriptor post-processing helpers with the reference's names and signatures
(core/modules/utils/descriptor_util.py), backed by csrc/desc.hip (kernels K6/K10)."""
import torch

from ...._native import desc_sample, normalize_map, upsample_normalize


def _f32(t):
    return t if t.dtype == torch.float32 else t.to(torch.float32)


def _scale(scale_factor):
    return float(scale_factor.detach()) if torch.is_tensor(scale_factor) else float(scale_factor)


def normalize_descriptors(raw_descriptors, scale_factor=1.0, normalize=True):
    """L2 normalisation over dim 1 times scale (descriptor_util.py:21-28)."""
    if not normalize:
        raise NotImplementedError("einx: un-normalised descriptors are not used by EI-Nexus")
    return normalize_map(_f32(raw_descriptors).contiguous(), _scale(scale_factor))


def get_dense_descriptors(normalized_descriptors):
    """[B,C,H,W] -> [B,H*W,C] view (descriptor_util.py:40-47)."""
    B, C = normalized_descriptors.shape[:2]
    return normalized_descriptors.reshape(B, C, -1).permute(0, 2, 1)


def _pack_positions(positions, width):
    """list of [n_i,>=2] (y,x) positions -> (indices [B,cap] int32, counts [B] int32); integer
    index arithmetic only (floor, y*W+x)."""
    B = len(positions)
    cap = max([int(p.shape[0]) for p in positions] + [1])
    dev = positions[0].device
    idx = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    cnt = torch.tensor([int(p.shape[0]) for p in positions], dtype=torch.int32, device=dev)
    for b, p in enumerate(positions):
        if p.shape[0]:
            yx = p[:, :2].floor().to(torch.int32)
            idx[b, :p.shape[0]] = yx[:, 0] * width + yx[:, 1]
    return idx, cnt


def sparsify_full_resolution_descriptors(raw_descriptors, positions, scale_factor=1.0, normalize=True):
    """integer gather + normalise for cell-1 networks (descriptor_util.py:50-71)."""
    if not normalize:
        raise NotImplementedError
    raw = _f32(raw_descriptors).contiguous()
    H, W = raw.shape[-2:]
    idx, cnt = _pack_positions(positions, W)
    out = desc_sample(raw, idx, cnt, (H, W), bilinear=False, scale=_scale(scale_factor))
    return tuple(out[b, :int(positions[b].shape[0])] for b in range(len(positions)))


def sparsify_low_resolution_descriptors(raw_descriptors, positions, image_size, scale_factor=1.0, normalize=True):
    """bilinear grid_sample at keypoints + normalise for cell-8 networks (descriptor_util.py:74-128)."""
    if not normalize:
        raise NotImplementedError
    raw = _f32(raw_descriptors).contiguous()
    Hp, Wp = int(image_size[0]), int(image_size[1])
    idx, cnt = _pack_positions(positions, Wp)
    out = desc_sample(raw, idx, cnt, (Hp, Wp), bilinear=True, scale=_scale(scale_factor))
    return [out[b, :int(positions[b].shape[0])] for b in range(len(positions))]


def upsample_descriptors(raw_descriptors, image_size, scale_factor=1.0):
    """bilinear resize + normalise (descriptor_util.py:131-138)."""
    return upsample_normalize(_f32(raw_descriptors).contiguous(), image_size, (0, 0, 0, 0), _scale(scale_factor))
