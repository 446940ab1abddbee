"""Padder (reference core/modules/utils/util.py:5-66).  In the native extractors the padding is
folded into the first convolution's addressing; this class is kept for API compatibility and for
callers that pad tensors themselves.  Pure copies / index shifts, no arithmetic."""
import torch
import torch.nn.functional as F

from ...._native import padder_pads


class Padder:
    def __init__(self, shape, p):
        self.shape = shape
        self.p = p
        h, w = shape[-2:]
        self.padding_size = padder_pads(h, w, p)  # (w0, w1, h0, h1)

    def pad(self, *args):
        return [F.pad(a, self.padding_size, mode="constant" if a.dtype == torch.bool else "replicate") for a in args]

    def unpad(self, *args):
        w0, w1, h0, h1 = self.padding_size
        out = []
        for a in args:
            h, w = a.shape[-2:]
            out.append(a[..., h0:h - h1, w0:w - w1].clone().contiguous())
        return out

    def unpad_positions(self, positions_list, ordering="xy"):
        assert ordering in ("xy", "yx")
        w0, _, h0, _ = self.padding_size
        first, second = (w0, h0) if ordering == "xy" else (h0, w0)
        out = []
        for p in positions_list:
            q = p.clone()
            q[..., 0] = p[..., 0] - first
            q[..., 1] = p[..., 1] - second
            out.append(q)
        return out
