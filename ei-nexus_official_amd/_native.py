"""Torch-tensor front door of the C ABI: validates device/dtype/contiguity, allocates outputs
and workspaces with torch (caching allocator, stream ordered) and enqueues the HIP kernels on the
current stream.  No arithmetic happens here; PyTorch is memory + stream plumbing only.
"""
import ctypes
import threading
import os
import math

import numpy as np
import torch

from . import _lib
from ._lib import ConvDesc, DetectParams, check

F32 = torch.float32


def lib():
    return _lib.load()


I32, I64, U8 = torch.int32, torch.int64, torch.uint8


def _dev_check(*tensors, dt=F32):
    """Every tensor must be a contiguous HIP tensor of dtype `dt`: the kernels read raw pointers and
    assume the element size (an fp16/fp64/int64 tensor would be read out of bounds or as garbage)."""
    for t in tensors:
        if t is None:
            continue
        if t.device.type != "cuda":
            raise RuntimeError(
                "einx: tensors must live on a HIP device (torch device type 'cuda'); there is no CPU path. "
                f"Got a tensor on {t.device}.")
        if t.dtype != dt:
            raise TypeError(f"einx: expected a {dt} tensor, got {t.dtype} (the kernels compute in fp32 with int32 counts/indices; "
                            "cast at the call site, e.g. x.float())")
        if not t.is_contiguous():
            raise RuntimeError("einx: tensors must be contiguous")


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _first_device(obj, depth=0):
    """HIP device of the first tensor found in a call argument (tensor, feats dict / list of tensors, PairBatch, BatchedFeats)."""
    if torch.is_tensor(obj):
        return obj.device if obj.device.type == "cuda" else None
    if depth > 2 or obj is None or isinstance(obj, (str, int, float, bool)):
        return None
    if isinstance(obj, dict):
        obj = list(obj.values())
    if isinstance(obj, (list, tuple)):
        for v in obj[:16]:
            d = _first_device(v, depth + 1)
            if d is not None:
                return d
        return None
    for name in ("desc", "score", "kpts"):  # PairBatch / BatchedFeats
        t = getattr(obj, name, None)
        if torch.is_tensor(t) and t.device.type == "cuda":
            return t.device
    return None


_scope = threading.local()


def on_input_device(fn):
    """The kernels are launched through ctypes on the stream of the INPUTS' device; HIP wants that device to be the current
    one.  torch's own operators switch devices per call, the reference therefore works with a model on cuda:1 while
    cuda:0 is current -- this decorator gives the drop-in's entry points the same behaviour (no-op on the current device).
    Entry points call each other (EIM.forward -> extract_batched -> ...): only the outermost one looks for the device, the
    nested ones run under its decision (their tensors derive from its inputs) -- the scan was 60 us of a 0.8 ms single-pair
    forward (round 5)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        if getattr(_scope, "depth", 0):
            return fn(*args, **kwargs)
        dev = None
        for a in args:
            if type(a) is torch.Tensor:  # the common case, without the generic walk
                if a.device.type == "cuda":
                    dev = a.device
                    break
        if dev is None:
            for a in list(args) + list(kwargs.values()):  # `self` (a module) holds no tensor attribute the scan looks at
                dev = _first_device(a)
                if dev is not None:
                    break
        if dev is None:
            return fn(*args, **kwargs)
        _scope.depth = 1
        try:
            if dev.index is None or dev.index == torch.cuda.current_device():
                return fn(*args, **kwargs)
            with torch.cuda.device(dev):
                return fn(*args, **kwargs)
        finally:
            _scope.depth = 0

    return wrapped


def _stream(t):
    # (torch.cuda.current_stream builds a Stream object per call: 6-7 us, nine times per forward)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(t.device.index if t.device.index is not None else torch.cuda.current_device()))


def padder_pads(h, w, p):
    """Padder.__init__ arithmetic (reference core/modules/utils/util.py:6-15) -> (w0, w1, h0, h1)."""
    hp = (((h // p) + 1) * p - h) % p
    wp = (((w // p) + 1) * p - w) % p
    return (wp // 2, wp - wp // 2, hp // 2, hp - hp // 2)


def topk_ranks(N, k):
    """Indices torch.quantile(q,'midpoint') gathers, in the reference's fp32 arithmetic."""
    q = np.float32(N - k) / np.float32(N)
    rank = np.float32(q * np.float32(N - 1))
    return int(math.floor(float(rank))), int(math.ceil(float(rank)))


def topk_capacity(N, top_k, det_thr):
    if not top_k or top_k >= N or det_thr < 1.0:
        return N
    lo, _ = topk_ranks(N, int(top_k))
    return N - 1 - lo


# ------------------------------------------------------------------------------ conv
class ConvLayer:
    """Kernel-native image of one conv block: weights repacked on device, BN(eval) folded into a
    per-channel (scale, shift) by einx_bn_fold (IEEE fp32, same op order as the oracle)."""

    def __init__(self, weight, bias, bn=None, relu=True, pool=False):
        _dev_check(weight)
        cout, cin, ks, _ = weight.shape
        self.cin, self.cout, self.ks, self.relu, self.pool = cin, cout, ks, relu, pool
        L = lib()
        w = weight.detach().to(F32).contiguous()
        self.w_native = torch.empty(L.einx_conv_weight_elems(cin, cout, ks), dtype=F32, device=w.device)
        check(L.einx_conv_repack(_ptr(w), cin, cout, ks, _ptr(self.w_native), _stream(w)), "einx_conv_repack")
        self.bias = None if bias is None else bias.detach().to(F32).contiguous()
        self.scale = self.shift = None
        if bn is not None:
            g, b, mean, var = [t.detach().to(w.device, F32).contiguous() for t in bn[:4]]
            self.scale = torch.empty(cout, dtype=F32, device=w.device)
            self.shift = torch.empty(cout, dtype=F32, device=w.device)
            check(L.einx_bn_fold(_ptr(g), _ptr(b), _ptr(mean), _ptr(var), float(bn[4]), cout, _ptr(self.scale), _ptr(self.shift),
                                 _stream(w)), "einx_bn_fold")
            self._keep_bn = (g, b, mean, var)
        self.desc = ConvDesc(self.w_native.data_ptr(), 0 if self.bias is None else self.bias.data_ptr(),
                             0 if self.scale is None else self.scale.data_ptr(), 0 if self.shift is None else self.shift.data_ptr(),
                             cin, cout, ks, int(relu), int(pool))
        self._keep = w  # repack reads it asynchronously

    def __call__(self, x, fold=None):
        """x [B,cin,Hs,Ws]; fold = (h0, w0, H, W) applies replicate padding on the fly."""
        _dev_check(x)
        B, C, Hs, Ws = x.shape
        if C != self.cin:
            raise ValueError(f"conv expects {self.cin} input channels, got {C}")
        h0, w0, H, W = fold if fold is not None else (0, 0, Hs, Ws)
        Ho, Wo = (H // 2, W // 2) if self.pool else (H, W)
        out = torch.empty((B, self.cout, Ho, Wo), dtype=F32, device=x.device)
        check(lib().einx_conv_block(_ptr(x), B, Hs, Ws, h0, w0, H, W, ctypes.byref(self.desc), _ptr(out), _stream(x)), "einx_conv_block")
        return out


def bn_tuple(bn):
    return (bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)


def div_inplace(x, divisor):
    _dev_check(x)
    check(lib().einx_div_inplace(_ptr(x), x.numel(), float(divisor), _stream(x)), "einx_div_inplace")
    return x


def image_prepare(image, divisor):
    """SuperPointv1's input handling for RGB / non-contiguous images (superpoint_extractor.py:372-376): `image /= divisor` in
    place THROUGH the tensor's strides, and the network's contiguous single-channel input (C == 3: kornia's rgb_to_grayscale of
    the scaled image) as a new tensor.  divisor 1.0 = the image has been scaled by an earlier call (x / 1 == x)."""
    if image.device.type != "cuda":
        raise RuntimeError(f"einx: input must be on a HIP device (no CPU path), got {image.device}")
    if image.dtype != F32:
        raise TypeError("einx extractors compute in fp32; pass a float32 tensor")
    B, C, H, W = image.shape  # any strides: the kernel walks them
    gray = torch.empty((B, 1, H, W), dtype=F32, device=image.device)
    if gray.numel():
        sb, sc, sh, sw = image.stride()
        check(lib().einx_image_prepare(_ptr(image), B, C, H, W, sb, sc, sh, sw, float(divisor), _ptr(gray), _stream(image)), "einx_image_prepare")
    return gray


# ------------------------------------------------------------------------------ detector
def score_map(logits, mask=None, pads=(0, 0, 0, 0), dilate=False, border=0):
    """-> (probability [B,C,hc,wc], score [B,1,Hp,Wp])."""
    _dev_check(logits)
    if mask is not None and mask.device.type != "cuda":
        raise RuntimeError(f"einx: the mask must live on a HIP device, got {mask.device}")
    B, C, hc, wc = logits.shape
    cell = 8 if C == 65 else 1
    Hp, Wp = hc * cell, wc * cell
    w0, w1, h0, h1 = pads
    H, W = Hp - h0 - h1, Wp - w0 - w1
    prob = torch.empty_like(logits)
    score = torch.empty((B, 1, Hp, Wp), dtype=F32, device=logits.device)
    m8 = None
    if mask is not None:
        if tuple(mask.shape[-2:]) != (H, W):
            raise ValueError(f"mask spatial size {tuple(mask.shape[-2:])} does not match the image ({H},{W})")
        # non-bool masks: non-zero is visible (the reference tests `conv(mask.float()) > 0`), see _extract._mask_u8
        m8 = (mask if mask.dtype == torch.bool else mask.ne(0)).contiguous().view(torch.uint8)
    check(lib().einx_score_map(_ptr(logits), B, C, hc, wc, _ptr(m8), H, W, h0, w0, int(dilate), int(border), _ptr(prob), _ptr(score),
                               _stream(logits)), "einx_score_map")
    return prob, score


def dense_positions(score, ordering="yx"):
    """score [B,1,H,W] (unpadded) -> [B,H*W,3] dense keypoint list (get_dense_positions)."""
    _dev_check(score)
    B, _, H, W = score.shape
    out = torch.empty((B, H * W, 3), dtype=F32, device=score.device)
    check(lib().einx_dense_positions(_ptr(score), B, H, W, int(ordering == "xy"), _ptr(out), _stream(score)), "einx_dense_positions")
    return out


def remove_border(score, border):
    _dev_check(score)
    B = score.shape[0]
    Hp, Wp = score.shape[-2:]
    check(lib().einx_remove_border(_ptr(score), int(score.numel() // (Hp * Wp)), Hp, Wp, int(border), _stream(score)), "einx_remove_border")
    return score


class Detection:
    """Device-side result of einx_detect for a batch."""
    __slots__ = ("positions", "indices", "counts", "thr", "not_converged", "nms", "cap", "padded", "pads", "stale")


def detect(score, *, top_k, radius, det_thr, pads=(0, 0, 0, 0), ordering="yx", cap=None, nms_iters=8, want_nms=True):
    """score [B,1,Hp,Wp] or [B,Hp,Wp] (already masked/border-zeroed)."""
    _dev_check(score)
    B = score.shape[0]
    Hp, Wp = score.shape[-2:]
    w0, w1, h0, h1 = pads
    H, W = Hp - h0 - h1, Wp - w0 - w1
    N = Hp * Wp
    if cap is None:
        cap = topk_capacity(N, top_k, det_thr)
    cap = max(int(cap), 1)
    p = DetectParams(B, Hp, Wp, H, W, h0, w0, int(radius), int(top_k or 0), float(det_thr), int(ordering == "xy"), cap, int(nms_iters))
    L = lib()
    dev = score.device
    ws = torch.empty(L.einx_detect_ws_bytes(ctypes.byref(p)), dtype=torch.uint8, device=dev)
    d = Detection()
    d.positions = torch.empty((B, cap, 3), dtype=F32, device=dev)
    d.indices = torch.empty((B, cap), dtype=torch.int32, device=dev)
    d.counts = torch.empty((B,), dtype=torch.int32, device=dev)
    d.thr = torch.empty((B,), dtype=F32, device=dev)
    d.not_converged = torch.empty((B,), dtype=torch.int32, device=dev)
    d.nms = torch.empty((B, H, W), dtype=F32, device=dev) if want_nms else None
    d.cap, d.padded, d.pads = cap, (Hp, Wp), pads
    check(L.einx_detect(_ptr(score), ctypes.byref(p), _ptr(ws), _ptr(d.nms), _ptr(d.positions), _ptr(d.indices), _ptr(d.counts),
                        _ptr(d.thr), _ptr(d.not_converged), _stream(score)), "einx_detect")
    return d


# ------------------------------------------------------------------------------ descriptors
def desc_sample(raw, indices, counts, padded_size, bilinear, scale, raw_cl=None):
    """raw [B,D,hc,wc].  raw_cl: the channels-last copy [B,hc*wc,D] from normalize_map(want_cl=True);
    when given (bilinear only) the taps are read from it -- same values, coalesced rows."""
    _dev_check(raw, raw_cl)
    _dev_check(indices, counts, dt=I32)
    B, D, hc, wc = raw.shape
    cap = indices.shape[1]
    out = torch.empty((B, cap, D), dtype=F32, device=raw.device)
    use_cl = raw_cl is not None and bilinear
    src = raw_cl if use_cl else raw
    check(lib().einx_desc_sample(_ptr(src), B, D, hc, wc, int(padded_size[0]), int(padded_size[1]), int(bilinear), int(use_cl),
                                 _ptr(indices), _ptr(counts), cap, float(scale), _ptr(out), _stream(raw)), "einx_desc_sample")
    return out


def normalize_map(raw, scale, want_cl=False):
    """-> normalised map like `raw`; with want_cl also the channels-last copy [B,P,D] of the raw values."""
    _dev_check(raw)
    B, D = raw.shape[:2]
    P = int(raw.numel() // (B * D))
    out = torch.empty_like(raw)
    cl = torch.empty((B, P, D), dtype=F32, device=raw.device) if (want_cl and D <= 512) else None
    check(lib().einx_normalize_map(_ptr(raw), B, D, P, float(scale), _ptr(out), _ptr(cl), _stream(raw)), "einx_normalize_map")
    return (out, cl) if want_cl else out


def upsample_normalize(raw, padded_size, pads, scale):
    _dev_check(raw)
    B, D, hc, wc = raw.shape
    Hp, Wp = int(padded_size[0]), int(padded_size[1])
    w0, w1, h0, h1 = pads
    H, W = Hp - h0 - h1, Wp - w0 - w1
    out = torch.empty((B, D, H, W), dtype=F32, device=raw.device)
    nws = int(lib().einx_upsample_ws_bytes(B, H, W))
    ws = torch.empty((nws + 3) // 4, dtype=F32, device=raw.device)  # per-pixel norms handed from the first kernel to the second
    check(lib().einx_upsample_normalize(_ptr(raw), B, D, hc, wc, Hp, Wp, h0, w0, H, W, float(scale), _ptr(out), _ptr(ws), nws, _stream(raw)),
          "einx_upsample_normalize")
    return out


# ------------------------------------------------------------------------------ matching
class MatchResult:
    __slots__ = ("matches0", "matches1", "scores0", "scores1", "la", "mk0", "mk1", "nmatch", "ref0", "ref1", "mk0_flat", "mk1_flat", "stale")

    def __init__(self):
        self.mk0_flat = self.mk1_flat = None
        self.stale = None  # device int32 [1]: the matcher's weight watch (LightGlue), read back with the match counts


def mnn(desc0, n, desc1, m, want_la=True, ratio_thresh=None, distance_thresh=None, gather=None):
    """ratio_thresh / distance_thresh: find_nn's optional thresholds (MNN.py:12-22), falsy = off.
    gather = (kpts0, kpts1, cols): also the matched keypoints of every pair (gather_matches' r.mk0 / r.mk1 / r.nmatch), in the
    same call -- without thresholds the mutual check and the compaction are one launch."""
    _dev_check(desc0, desc1)
    _dev_check(n, m, dt=I32)
    B, cap0, D = desc0.shape
    cap1 = desc1.shape[1]
    L = lib()
    dev = desc0.device
    ws = torch.empty(L.einx_mnn_ws_bytes(B, cap0, cap1), dtype=torch.uint8, device=dev)
    r = MatchResult()
    r.matches0 = torch.empty((B, cap0), dtype=torch.int64, device=dev)
    r.matches1 = torch.empty((B, cap1), dtype=torch.int64, device=dev)
    r.scores0 = torch.empty((B, cap0), dtype=F32, device=dev)
    r.scores1 = torch.empty((B, cap1), dtype=F32, device=dev)
    r.la = torch.empty((B, cap0 + 1, cap1 + 1), dtype=F32, device=dev) if want_la else None
    r.ref0 = r.ref1 = None
    if ratio_thresh or distance_thresh:
        # the squared Python scalars are rounded to fp32 once, as torch does for `tensor <= scalar` / `scalar * tensor`
        r2 = float(np.float32(float(ratio_thresh) ** 2)) if ratio_thresh else 0.0
        t2 = float(np.float32(float(distance_thresh) ** 2)) if distance_thresh else 0.0
        check(L.einx_mnn_thresh(_ptr(desc0), _ptr(n), cap0, _ptr(desc1), _ptr(m), cap1, B, D, int(bool(ratio_thresh)), r2,
                                int(bool(distance_thresh)), t2, _ptr(ws), _ptr(r.matches0), _ptr(r.matches1), _ptr(r.scores0), _ptr(r.scores1),
                                _ptr(r.la), _stream(desc0)), "einx_mnn_thresh")
        return r if gather is None else gather_matches(r, gather[0], gather[1], n, gather[2])
    if gather is not None:
        k0, k1, cols = gather
        _dev_check(k0, k1)
        r.mk0 = torch.empty((B, cap0, cols), dtype=F32, device=dev)
        r.mk1 = torch.empty((B, cap0, cols), dtype=F32, device=dev)
        r.nmatch = torch.empty((B,), dtype=torch.int32, device=dev)
        check(L.einx_mnn_gather(_ptr(desc0), _ptr(n), cap0, _ptr(desc1), _ptr(m), cap1, B, D, _ptr(ws), _ptr(r.matches0), _ptr(r.matches1),
                                _ptr(r.scores0), _ptr(r.scores1), _ptr(r.la), _ptr(k0), _ptr(k1), int(cols), _ptr(r.mk0), _ptr(r.mk1),
                                _ptr(r.nmatch), _stream(desc0)), "einx_mnn_gather")
        return r
    check(L.einx_mnn(_ptr(desc0), _ptr(n), cap0, _ptr(desc1), _ptr(m), cap1, B, D, _ptr(ws), _ptr(r.matches0), _ptr(r.matches1),
                     _ptr(r.scores0), _ptr(r.scores1), _ptr(r.la), _stream(desc0)), "einx_mnn")
    return r


def gather_matches(r, kpts0, kpts1, n, cols):
    _dev_check(kpts0, kpts1)
    _dev_check(n, dt=I32)
    _dev_check(r.matches0, dt=I64)
    B, cap0, _ = kpts0.shape
    cap1 = kpts1.shape[1]
    dev = kpts0.device
    r.mk0 = torch.empty((B, cap0, cols), dtype=F32, device=dev)
    r.mk1 = torch.empty((B, cap0, cols), dtype=F32, device=dev)
    r.nmatch = torch.empty((B,), dtype=torch.int32, device=dev)
    check(lib().einx_gather_matches(_ptr(kpts0), _ptr(kpts1), _ptr(r.matches0), _ptr(n), cap0, cap1, B, cols, _ptr(r.mk0), _ptr(r.mk1),
                                    _ptr(r.nmatch), _stream(kpts0)), "einx_gather_matches")
    return r


def compact_matches(r):
    """r.mk0/r.mk1 [B,cap0,cols] + r.nmatch -> r.mk0_flat/r.mk1_flat [B*cap0,cols] packed pair after pair."""
    B, cap0, cols = r.mk0.shape
    if B == 1:  # one pair: its matched rows already sit at the front of mk0[0] / mk1[0] (no launch on the single-pair chain)
        r.mk0_flat, r.mk1_flat = r.mk0[0], r.mk1[0]
        return r
    r.mk0_flat = torch.empty((B * cap0, cols), dtype=F32, device=r.mk0.device)
    r.mk1_flat = torch.empty((B * cap0, cols), dtype=F32, device=r.mk0.device)
    check(lib().einx_compact_rows(_ptr(r.mk0), _ptr(r.mk1), _ptr(r.nmatch), B, cap0, cols, _ptr(r.mk0_flat), _ptr(r.mk1_flat),
                                  _stream(r.mk0)), "einx_compact_rows")
    return r


def lightglue(weights, pb0, pb1, want_la=True, want_ref=False, all_layers=False):
    """weights: _lib.LgWeights; pb0/pb1: PairBatch (kpts [B,cap,3], desc [B,cap,Din], counts).
    all_layers: ref0/ref1 are [B,n_layers,cap,d] (training-mode ref_descriptors) instead of [B,cap,d]."""
    _dev_check(pb0.kpts, pb0.desc, pb1.kpts, pb1.desc)
    _dev_check(pb0.counts, pb1.counts, dt=I32)
    B, cap0, cap1 = pb0.B, pb0.cap, pb1.cap
    L = lib()
    dev = pb0.desc.device
    d = int(weights.d)
    nbytes = L.einx_lg_ws_bytes_heads(B, cap0, cap1, d, int(weights.heads), int(weights.input_dim))
    if not nbytes:
        raise NotImplementedError("einx LightGlue: descriptor_dim must be num_heads x head_dim with head_dim a multiple of 4, at most 256")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    r = MatchResult()
    r.matches0 = torch.empty((B, cap0), dtype=torch.int64, device=dev)
    r.matches1 = torch.empty((B, cap1), dtype=torch.int64, device=dev)
    r.scores0 = torch.empty((B, cap0), dtype=F32, device=dev)
    r.scores1 = torch.empty((B, cap1), dtype=F32, device=dev)
    r.la = torch.empty((B, cap0 + 1, cap1 + 1), dtype=F32, device=dev) if want_la else None
    nl = int(weights.n_layers)
    if want_ref and all_layers:
        r.ref0 = torch.zeros((B, nl, cap0, d), dtype=F32, device=dev)
        r.ref1 = torch.zeros((B, nl, cap1, d), dtype=F32, device=dev)
    else:
        r.ref0 = torch.empty((B, cap0, d), dtype=F32, device=dev) if want_ref else None
        r.ref1 = torch.empty((B, cap1, d), dtype=F32, device=dev) if want_ref else None
    (h0, w0), (h1, w1) = pb0.image_size, pb1.image_size
    check(L.einx_lightglue(ctypes.byref(weights), _ptr(pb0.kpts), _ptr(pb0.desc), _ptr(pb0.counts), cap0, _ptr(pb1.kpts), _ptr(pb1.desc),
                           _ptr(pb1.counts), cap1, B, float(h0), float(w0), float(h1), float(w1), _ptr(ws), _ptr(r.matches0),
                           _ptr(r.matches1), _ptr(r.scores0), _ptr(r.scores1), _ptr(r.la), _ptr(r.ref0), _ptr(r.ref1),
                           nl if (want_ref and all_layers) else 1, _stream(pb0.desc)),
          "einx_lightglue")
    return r


def similarity(desc0, n, desc1, m):
    """[B,cap0,cap1] descriptor similarity (MNN.py:88); zero outside the first n[b] x m[b] block."""
    _dev_check(desc0, desc1)
    _dev_check(n, m, dt=I32)
    B, cap0, D = desc0.shape
    cap1 = desc1.shape[1]
    sim = torch.empty((B, cap0, cap1), dtype=F32, device=desc0.device)
    check(lib().einx_similarity(_ptr(desc0), _ptr(n), cap0, _ptr(desc1), _ptr(m), cap1, B, D, _ptr(sim), _stream(desc0)), "einx_similarity")
    return sim


def normalize_rows(x, scale):
    """F.normalize(x, dim=1) * scale for a [R,C] matrix."""
    _dev_check(x)
    R, C = x.shape
    out = torch.empty_like(x)
    if R:
        check(lib().einx_normalize_rows(_ptr(x), R, C, float(scale), _ptr(out), _stream(x)), "einx_normalize_rows")
    return out


def normalize_keypoints(kpts, size, out_cols=2):
    """lightglue.py:137-148 on [..., cols>=2] keypoints -> [..., out_cols] (zeros beyond column 1)."""
    _dev_check(kpts)
    cols = kpts.shape[-1]
    rows = kpts.numel() // cols
    out = torch.empty(tuple(kpts.shape[:-1]) + (out_cols,), dtype=F32, device=kpts.device)
    if rows:
        check(lib().einx_normalize_keypoints(_ptr(kpts), rows, cols, float(size[0]), float(size[1]), _ptr(out), out_cols, _stream(kpts)),
              "einx_normalize_keypoints")
    return out


def random_positions(u, size):
    """u [R,2] uniform draws -> [R,3] = (u0*size0, u1*size1, 0)  (Matchers.py:80-91)."""
    _dev_check(u)
    R = u.shape[0]
    out = torch.empty((R, 3), dtype=F32, device=u.device)
    if R:
        check(lib().einx_random_positions(_ptr(u), R, float(size[0]), float(size[1]), _ptr(out), _stream(u)), "einx_random_positions")
    return out


def linear(x, w, bias, out=None, accumulate=False):
    """y = x @ w.T + bias on the fp32 matrix cores (einx_linear); accumulate: y += ..."""
    _dev_check(x, w, bias, out)
    M, K = x.shape
    Nn = w.shape[0]
    y = out if out is not None else torch.empty((M, Nn), dtype=F32, device=x.device)
    check(lib().einx_linear(_ptr(x), M, K, _ptr(w), _ptr(bias), Nn, _ptr(y), int(accumulate), _stream(x)), "einx_linear")
    return y


WATCH_CHUNK_WORDS = 4096  # include/einx.h: EINX_WATCH_CHUNK_WORDS


class ParamWatch:
    """Device-side content watch of a module's fp32 parameters / buffers (einx_params_hash): `.data` edits, which no host-side
    version counter sees, raise `stale` at the next forward.  Exact since round 5: EVERY word is hashed (round 4 sampled 65 per
    tensor), tensors cut into rows of 4096 words, one wave per row -- 1.3 M extractor weights are 380 rows riding on spare
    workgroups of the descriptor sampling kernel, LightGlue's 12 M one 10 us launch.  Built when the module packs its weights;
    `check()` enqueues one small launch on the current stream; the flag travels to the host with the counts the forward reads
    back anyway."""

    def __init__(self, tensors):
        ts = [t.detach() for t in tensors if torch.is_tensor(t) and t.dtype == F32 and t.numel() > 0 and t.device.type == "cuda"
              and t.is_contiguous()]
        if os.environ.get("EINX_NO_WATCH") == "1":  # tools: A/B of the watch's cost
            ts = []
        self.keep = ts
        self.stale = None
        rows = [[t.data_ptr() + 4 * off, min(WATCH_CHUNK_WORDS, t.numel() - off)] for t in ts for off in range(0, t.numel(), WATCH_CHUNK_WORDS)]
        self.n = len(rows)
        if not rows:
            return
        dev = ts[0].device
        self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.ref = torch.empty((self.n,), dtype=torch.int64, device=dev)
        self.scratch = torch.empty((self.n,), dtype=torch.int64, device=dev)
        self.stale = torch.zeros((1,), dtype=torch.int32, device=dev)
        check(lib().einx_params_hash(_ptr(self.table), self.n, _ptr(self.ref), None, None, _stream(self.table)), "einx_params_hash")

    def check(self):
        """enqueue the comparison on the current stream; returns the device flag (int32 [1], 1 = some tensor changed)"""
        if self.n:
            check(lib().einx_params_hash(_ptr(self.table), self.n, _ptr(self.scratch), _ptr(self.ref), _ptr(self.stale), _stream(self.table)),
                  "einx_params_hash")
        return self.stale
