"""Event representations on the GPU, same names and argument meaning as the
reference's datasets/representations.py:67-124 (`events_to_voxel_grid`) and the events mask built in
datasets/visualize.py:23-50 + test_events-image_same-time.py:137.  Events arrive as the reference's
dict of numpy arrays {"x","y","t","p"}; the result stays on the device, ready for EIM.forward."""
import ctypes

import numpy as np
import torch

from .. import _native as N
from .._lib import check


def _pack(events_list, device):
    xs, ys, ts, ps, offs = [], [], [], [], [0]
    for ev in events_list:
        xs.append(np.asarray(ev["x"], np.float32))
        ys.append(np.asarray(ev["y"], np.float32))
        ts.append(np.asarray(ev["t"], np.float64))
        ps.append(np.asarray(ev["p"], np.float32))
        offs.append(offs[-1] + len(xs[-1]))
    cat = lambda v, dt: torch.from_numpy(np.ascontiguousarray(np.concatenate(v).astype(dt))).to(device)  # noqa: E731
    return cat(xs, np.float32), cat(ys, np.float32), cat(ts, np.float64), cat(ps, np.float32), np.asarray(offs, np.int64)


def events_to_voxel_grid_batch(events_list, input_size, normalize=True, device="cuda", packed=None):
    """list of B event dicts -> voxel grids [B,bins,H,W] (fp32, on `device`).  packed: the result of `_pack` for these events
    (a caller that also needs the events mask packs and uploads the arrays once)."""
    bins, H, W = (int(v) for v in input_size)
    B = len(events_list)
    x, y, t, p, offs = packed if packed is not None else _pack(events_list, device)
    L = N.lib()
    grid = torch.empty((B, bins, H, W), dtype=torch.float32, device=device)
    ws = torch.empty(L.einx_voxel_ws_bytes(B, bins, H, W, int(offs[-1])), dtype=torch.uint8, device=device)
    check(L.einx_voxel_grid(N._ptr(x), N._ptr(y), N._ptr(t), N._ptr(p), offs.ctypes.data_as(ctypes.c_void_p), B, bins, H, W, int(normalize),
                            N._ptr(grid), N._ptr(ws), ws.numel(), N._stream(grid)), "einx_voxel_grid")
    return grid


def events_to_voxel_grid(events, input_size, normalize=True, device="cuda"):
    """Drop-in for datasets/representations.py:67-124 (one sample): returns [bins,H,W].
    Unlike the reference it does not modify the `events` dict in place."""
    return events_to_voxel_grid_batch([events], input_size, normalize, device)[0]


def events_mask_batch(events_list, resolution, device="cuda", packed=None):
    """`draw_events_accumulation_image(events, (W,H)) > 0` for B samples -> bool [B,1,H,W]."""
    W, H = (int(v) for v in resolution)
    B = len(events_list)
    x, y, _, _, offs = packed if packed is not None else _pack(events_list, device)
    L = N.lib()
    mask = torch.empty((B, 1, H, W), dtype=torch.uint8, device=device)
    ws = torch.empty(L.einx_events_ws_bytes(B, H, W), dtype=torch.uint8, device=device)
    check(L.einx_events_mask(N._ptr(x), N._ptr(y), offs.ctypes.data_as(ctypes.c_void_p), B, H, W, N._ptr(ws), N._ptr(mask), N._stream(mask)),
          "einx_events_mask")
    return mask.view(torch.bool)


def events_representation_batch(events_list, input_size, normalize=True, device="cuda"):
    """voxel grids [B,bins,H,W] and events masks [B,1,H,W] of B samples from ONE host-side packing and upload of the raw
    event arrays (what test_events-image_same-time.py:130-140 builds per sample with two passes over the events)."""
    bins, H, W = (int(v) for v in input_size)
    packed = _pack(events_list, device)
    return (events_to_voxel_grid_batch(events_list, input_size, normalize, device, packed=packed),
            events_mask_batch(events_list, (W, H), device, packed=packed))
