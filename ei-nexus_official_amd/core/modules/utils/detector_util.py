"""Detector post-processing helpers with the reference's names and signatures
(core/modules/utils/detector_util.py), each backed by the HIP kernels of csrc/detect.hip.
They are the unit-test entry points for kernels K3-K5 and keep the evaluation scripts' imports
working (test_events-image_same-time.py:31-37)."""
import torch

from ...._native import detect, remove_border, score_map


def _f32(t):
    """the kernels compute in fp32; other float dtypes (autocast halves, doubles), which the reference's
    torch ops would accept, are cast here instead of being misread through a raw pointer"""
    return t if t.dtype == torch.float32 else t.to(torch.float32)


def logits_to_prob(logits, channel_dim=1):
    """softmax over 65 channels / sigmoid for 1 channel (detector_util.py:18-40)."""
    if channel_dim != 1 or logits.dim() != 4:
        raise NotImplementedError("einx: logits must be [B,C,h,w] with channel_dim=1")
    prob, _ = score_map(_f32(logits).contiguous())
    return prob


def depth_to_space(prob, cell_size=8, channel_dim=1):
    """drop the dustbin and pixel-shuffle (detector_util.py:43-77).  Pure re-indexing (views and a
    copy), so it is expressed with tensor reshapes; the fused production path is einx_score_map."""
    if cell_size > 1:
        assert prob.shape[channel_dim] == cell_size * cell_size + 1
        B, _, h, w = prob.shape
        p = prob[:, :cell_size * cell_size].reshape(B, cell_size, cell_size, h, w)
        return p.permute(0, 3, 1, 4, 2).reshape(B, 1, h * cell_size, w * cell_size)
    assert prob.shape[channel_dim] == 1
    return prob


def remove_border_points(image_nms, border_dist=4):
    """in place, like the reference (detector_util.py:138-164)."""
    if border_dist > 0:
        remove_border(image_nms, border_dist)
    return image_nms


def _neg_inf():
    return float("-inf")


def fast_nms(image_probs, nms_dist=4, max_iter=-1, min_value=0.0):
    """fix-point NMS (detector_util.py:243-337).  max_iter/min_value other than the defaults are
    not used anywhere in EI-Nexus and are not implemented."""
    if nms_dist == 0:
        return image_probs
    if max_iter != -1 or min_value != 0.0:
        raise NotImplementedError
    shape = image_probs.shape
    m = _f32(image_probs).reshape(-1, shape[-2], shape[-1]).contiguous()
    iters = 8
    while True:
        d = detect(m, top_k=0, radius=nms_dist, det_thr=_neg_inf(), cap=1, nms_iters=iters)
        if not bool(d.not_converged.any()):
            return d.nms.reshape(shape)
        iters *= 4


def prob_map_to_points_map(prob_map, prob_thresh=0.015, nms_dist=4, border_dist=4, use_fast_nms=True, top_k=None):
    """border removal (in place on prob_map) -> NMS -> top-k / threshold (detector_util.py:80-135)."""
    if not use_fast_nms:
        raise NotImplementedError("einx implements the fast_nms path the extractors use")
    remove_border_points(prob_map, border_dist)  # in place: needs an fp32 map (TypeError otherwise)
    m = prob_map.squeeze(1) if prob_map.dim() == 4 else prob_map
    m = m.contiguous()
    iters = 8
    while True:
        d = detect(m, top_k=int(top_k or 0), radius=nms_dist, det_thr=float(prob_thresh), cap=1, nms_iters=iters)
        if not bool(d.not_converged.any()):
            return d.nms
        iters *= 4


def prob_map_to_positions_with_prob(prob_map, threshold=0.0, ordering="yx"):
    """raster-order nonzero + 0.5, with the probability as third column (detector_util.py:451-484)."""
    m = prob_map.squeeze(1) if prob_map.dim() == 4 else prob_map
    m = _f32(m).contiguous()
    B, H, W = m.shape
    d = detect(m, top_k=0, radius=0, det_thr=float(threshold), ordering=ordering, cap=H * W, want_nms=False)
    counts = d.counts.cpu().tolist()
    return tuple(d.positions[b, :counts[b]] for b in range(B))


def get_dense_positions(probability, ordering="yx"):
    """meshgrid + 0.5 with the probability appended (detector_util.py:504-519); index plumbing."""
    B, _, H, W = probability.shape
    dev = probability.device
    ys = torch.arange(H, device=dev, dtype=torch.float32) + 0.5
    xs = torch.arange(W, device=dev, dtype=torch.float32) + 0.5
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    first, second = (gy, gx) if ordering == "yx" else (gx, gy)
    grid = torch.stack([first, second], -1).reshape(1, -1, 2).expand(B, -1, -1)
    return torch.cat((grid, probability.reshape(B, -1, 1)), dim=2)
