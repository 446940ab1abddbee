"""Batched device-side evaluation metrics (csrc/metrics.hip)."""
import ctypes

import torch

from ..._native import on_input_device
from ... import _native as N
from ..._lib import MetricParams, check


def metric_names(mma_thr=(1, 3), vdd_thr=(1, 3), prefix_vdd="VDD"):
    names = ["MR"] + [f"MMA@{t}" for t in mma_thr]
    for t in vdd_thr:
        names += [f"{prefix_vdd}_Repeatability@{t}", f"{prefix_vdd}_ValidDistance@{t}", f"{prefix_vdd}_Angle@{t}"]
    return names


@on_input_device
def pair_metrics(kpts0, desc0, n, kpts1, desc1, m, mk0, mk1, nmatch, size0, size1, homography=None, mma_thr=(1, 3), vdd_thr=(1, 3),
                 ordering="yx", rep_nan_if_empty=False):
    """All tensors on the device: kpts [B,cap,3], desc [B,cap,D], counts int32 [B], matched keypoints
    [B,cap0,cols] + nmatch.  Returns float64 [B, 1+len(mma_thr)+3*len(vdd_thr)] (see metric_names)."""
    B, cap0, _ = kpts0.shape
    cap1 = kpts1.shape[1]
    p = MetricParams()
    p.B, p.cap0, p.cap1, p.D, p.cols = B, cap0, cap1, desc0.shape[-1], mk0.shape[-1]
    p.H0, p.W0, p.H1, p.W1 = int(size0[0]), int(size0[1]), int(size1[0]), int(size1[1])
    p.kp_yx = int(ordering == "yx")
    p.n_mma, p.n_vdd = len(mma_thr), len(vdd_thr)
    p.rep_nan_if_empty = int(bool(rep_nan_if_empty))
    for i, t in enumerate(mma_thr):
        p.mma_thr[i] = float(t)
    for i, t in enumerate(vdd_thr):
        p.vdd_thr[i] = float(t)
    L = N.lib()
    dev = kpts0.device
    hom = None
    if homography is not None:
        hom = homography.to(dev, torch.float32).reshape(B, 9).contiguous()
    ws = torch.empty(L.einx_metrics_ws_bytes(ctypes.byref(p)), dtype=torch.uint8, device=dev)
    out = torch.empty((B, 1 + p.n_mma + 3 * p.n_vdd), dtype=torch.float64, device=dev)
    N._dev_check(kpts0, kpts1, desc0, desc1, mk0, mk1)
    N._dev_check(n, m, nmatch, dt=torch.int32)
    check(L.einx_pair_metrics(ctypes.byref(p), N._ptr(kpts0), N._ptr(kpts1), N._ptr(desc0), N._ptr(desc1), N._ptr(n), N._ptr(m), N._ptr(mk0),
                              N._ptr(mk1), N._ptr(nmatch), N._ptr(hom), N._ptr(ws), N._ptr(out), N._stream(kpts0)), "einx_pair_metrics")
    return out


@on_input_device
def batch_metrics(ev, im, mr, homography=None, mma_thr=(1, 3), vdd_thr=(1, 3)):
    """Metrics for a whole EIM.forward_batched result (BatchedFeats x2 + MatchResult), no host sync."""
    return pair_metrics(ev.det.positions, ev.sparse_desc, ev.det.counts, im.det.positions, im.sparse_desc, im.det.counts, mr.mk0, mr.mk1,
                        mr.nmatch, ev.image_size, im.image_size, homography, mma_thr, vdd_thr, ordering=ev.ordering)


def _pad3(k):
    if k.shape[-1] == 3:
        return k
    return torch.cat([k, k.new_zeros(k.shape[0], 3 - k.shape[-1])], 1)


def single_pair(points1, points2, desc1, desc2, matched1, matched2, size0, size1, homography, mma_thr, vdd_thr, ordering="yx",
                rep_nan_if_empty=False):
    """update_one-style entry: per-pair tensors of any length -> dict of python floats."""
    dev = points1.device
    k0, k1 = _pad3(points1.float())[None].contiguous(), _pad3(points2.float())[None].contiguous()
    M = int(matched1.shape[0]) if matched1 is not None else 0
    cap0, cap1 = max(k0.shape[1], M, 1), max(k1.shape[1], 1)  # matched rows share image 0's capacity in the ABI
    D = desc1.shape[-1] if desc1 is not None else 4

    def fit(t, cap, width):
        out = torch.zeros((1, cap, width), dtype=torch.float32, device=dev)
        if t is not None and t.numel():
            out[0, :t.shape[0], :t.shape[1]] = t
        return out
    k0, k1 = fit(k0[0], cap0, 3), fit(k1[0], cap1, 3)
    d0, d1 = fit(desc1, cap0, D), fit(desc2, cap1, D)
    cols = matched1.shape[-1] if matched1 is not None and matched1.numel() else 3
    mk0, mk1 = fit(matched1, cap0, cols), fit(matched2, cap0, cols)
    cnt = lambda v: torch.tensor([v], dtype=torch.int32, device=dev)  # noqa: E731
    hom = None if homography is None else homography.reshape(1, 3, 3)
    out = pair_metrics(k0, d0, cnt(points1.shape[0]), k1, d1, cnt(points2.shape[0]), mk0, mk1, cnt(M), size0, size1, hom, mma_thr, vdd_thr,
                       ordering, rep_nan_if_empty=rep_nan_if_empty)
    return dict(zip(metric_names(mma_thr, vdd_thr), out[0].tolist()))
