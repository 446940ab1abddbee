"""Batched extractor engine shared by the four extractor front-ends (VGGExtractor,
VGGExtractorNP, SuperPointv1, SiLKModel).  Everything stays on the device and on the current
stream; nothing here synchronises with the host.  `BatchedFeats.materialize()` turns the device
result into the reference's output dict once the per-image keypoint counts are on the host.

Reference behaviour mirrored (file:line): core/modules/event_extractors/EventExtractors.py:517-624
and :331-434, core/modules/image_extractors/superpoint_extractor.py:345-480,
core/modules/image_extractors/silk_extractor.py:177-257.
"""
import ctypes
import os

import torch

from . import _lib
from . import _native as N


_SIZE_CACHE = {}


def _size_tensor(H, W, dev):
    """int64 [2] = (H, W) on the device (the reference's feats["image_size"] entries), cached: building
    it costs a host-to-device copy per call otherwise."""
    key = (H, W, str(dev))
    t = _SIZE_CACHE.get(key)
    if t is None:
        t = _SIZE_CACHE[key] = torch.tensor([H, W], device=dev)
    return t


class _Lazy:
    """value of a FeatsDict entry that is computed when it is first read"""
    __slots__ = ("fn",)

    def __init__(self, fn):
        self.fn = fn

    def __repr__(self):
        return "<einx: computed on first access>"


class FeatsDict(dict):
    """The reference's feature dict plus a handle on the device-side batch (`_batched`) so that the
    matcher can consume it without re-packing the per-image lists.

    Round 4 ("dense outputs on demand"): with `dense_outputs = "lazy"` (the default) the dense by-products --
    `normalized_descriptors` [B,C,H,W] (2.95 GB per side at B=32), `dense_descriptors`, `dense_positions` -- are dict entries
    from the start (same keys as the reference's dict) whose VALUES are computed by the same kernels when they are first
    read, on the stream current at that moment.  A caller that only reads the sparse outputs (the reference's evaluation
    scripts) never pays for them; a caller that reads them gets the tensors the eager mode would have produced, provided it
    has not modified `raw_descriptors` / `score` in place in between.  Every read path resolves them: d[k], get, items,
    values, pop, popitem, setdefault, copy, `d | other` / `other | d`, `==` (compared on the resolved values, as plain dicts
    would be), iteration-based copies (`dict(d)`, `{**d}`: __iter__ is overridden so that CPython's merge takes the generic
    keys() + __getitem__ route); `reversed(d)` yields keys like any dict."""
    _batched = None

    def _resolve(self, k, v):
        if isinstance(v, _Lazy):
            v = v.fn(self)  # the dict it is read from (EIM.forward_graph re-issues the same lazy entries in a fresh dict per call)
            dict.__setitem__(self, k, v)
        return v

    def __getitem__(self, k):
        return self._resolve(k, dict.__getitem__(self, k))

    def get(self, k, default=None):
        return self[k] if k in self else default

    def __iter__(self):
        return dict.__iter__(self)

    def items(self):
        return [(k, self[k]) for k in dict.keys(self)]

    def values(self):
        return [self[k] for k in dict.keys(self)]

    def pop(self, k, *default):
        if k in self:
            v = self[k]
            dict.__delitem__(self, k)
            return v
        if default:
            return default[0]
        raise KeyError(k)

    def setdefault(self, k, default=None):
        if k in self:
            return self[k]
        dict.__setitem__(self, k, default)
        return default

    def popitem(self):
        k = next(reversed(self))  # dict.popitem's LIFO order
        return k, self.pop(k)

    def _resolve_all(self):
        for k in dict.keys(self):
            self[k]
        return self

    def __or__(self, other):  # d | other, other | d, ==: plain-dict semantics on the resolved values
        return dict(self.items()) | (dict(other.items()) if isinstance(other, FeatsDict) else other)

    def __ror__(self, other):
        return (dict(other.items()) if isinstance(other, FeatsDict) else dict(other)) | dict(self.items())

    def __ior__(self, other):
        self.update(other.items() if isinstance(other, FeatsDict) else other)
        return self

    def __eq__(self, other):
        if isinstance(other, FeatsDict):
            other._resolve_all()
        return dict.__eq__(self._resolve_all(), other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def copy(self):
        out = FeatsDict(self.items())
        out._batched = self._batched
        return out

    def lazy_keys(self):
        """keys whose values have not been computed yet (tests / tools)"""
        return [k for k in dict.keys(self) if isinstance(dict.__getitem__(self, k), _Lazy)]


class BatchedFeats:
    def __init__(self):
        self.kind = None
        self.cell = 8
        self.B = 0
        self.image_size = None  # (H, W)
        self.pads = None  # (w0, w1, h0, h1)
        self.padded = None  # (Hp, Wp)
        self.feats = self.logits = self.raw = self.prob = self.score = None
        self.score_crop = None  # [B,1,H,W] written by einx_extract's score kernel (handle-level path); None: cropped from `score`
        self.det = None
        self.sparse_desc = None
        self.raw_cl = None
        self.coarse = None
        self.normalized = None
        self.scale = 1.0
        self.shift = 0.0  # padding=0 networks: keypoints are mapped back by +9 (mapping_positions)
        self.ordering = "yx"
        self.dense = False
        self.dense_lazy = False
        self._prepared = None
        self._full_lists = None
        self._ns = None

    # ------------------------------------------------------------------ device handles
    @property
    def counts(self):
        return self.det.counts

    @property
    def positions(self):
        return self.det.positions

    def run_dense(self):
        """The dense by-product (`normalized_descriptors`: upsample + normalise, cropped), enqueued on the CURRENT stream.  Part
        of the extractor call unless the caller deferred it (EIM schedules it beside the other extractor's convolutions)."""
        if not (self.dense or self.dense_lazy) or self.normalized is not None:
            return
        w0, w1, h0, h1 = self.pads
        Hp, Wp = self.padded
        if self.cell == 8:
            self.normalized = N.upsample_normalize(self.raw, (Hp, Wp), self.pads, self.scale)
        else:
            nd = N.normalize_map(self.raw, self.scale)
            self.normalized = nd[:, :, h0:Hp - h1, w0:Wp - w1].clone().contiguous()

    def prepare(self):
        """Everything of the output dict that does not depend on the keypoint counts.  Called BEFORE
        the host waits for the counts, so the crop/clone kernels and the Python work run while the
        device is still busy with the detection tail instead of leaving it idle afterwards."""
        if self._prepared is not None:
            return self._prepared
        H, W = self.image_size
        w0, w1, h0, h1 = self.pads
        Hp, Wp = self.padded
        dev = self.score.device
        size_t = _size_tensor(H, W, dev)
        out = FeatsDict()
        out["image_size"] = [size_t] * self.B
        out["backbone_feats"] = self.feats
        out["logits"] = self.logits
        out["raw_descriptors"] = self.raw
        out["probability"] = self.prob
        out["score"] = self.score_crop if self.score_crop is not None else self.score[:, :, h0:Hp - h1, w0:Wp - w1].clone().contiguous()
        out["nms"] = None  # filled by materialize (a re-detection replaces it)
        if self.cell == 8:
            out["coarse_descriptors"] = self.coarse
        def dense_descriptors(d):
            nd = d["normalized_descriptors"]
            return [nd[b].permute(1, 2, 0).reshape(-1, nd.shape[1]) for b in range(self.B)]

        def dense_positions(d):
            with torch.cuda.device(self.score.device):
                dp = N.dense_positions(dict.__getitem__(d, "score"), self.ordering)
                if self.shift:
                    dp[:, :, :2] += self.shift
            return list(dp.unbind(0))

        if self.dense:
            self.run_dense()  # no-op unless a caller deferred it and never ran it
            out["normalized_descriptors"] = self.normalized
            out["dense_descriptors"] = dense_descriptors(out)
            out["dense_positions"] = dense_positions(out)
        elif self.dense_lazy:
            def normalized(d):
                with torch.cuda.device(self.raw.device):
                    self.run_dense()
                return self.normalized

            for k, fn in (("normalized_descriptors", normalized), ("dense_descriptors", dense_descriptors), ("dense_positions", dense_positions)):
                dict.__setitem__(out, k, _Lazy(fn))
        # speculative per-image lists for the common case that every image fills its top-k quota
        # (checked against the real counts in materialize)
        self._full_lists = (self.det, list(self.sparse_desc.unbind(0)), list(self.det.positions.unbind(0)))
        self._prepared = out
        return out

    def prepared_template(self):
        """EIM.forward_graph: the count-independent part of the dict as built during capture (its tensors are graph-owned
        buffers refreshed by every replay); lazy entries stay lazy"""
        out = self.prepare()
        return [(k, dict.__getitem__(out, k)) for k in dict.keys(out)]

    def reuse_prepared(self, template):
        out = FeatsDict()
        for k, v in template:
            dict.__setitem__(out, k, v)
        self._prepared = out
        self.normalized = None if self.dense_lazy else self.normalized

    def materialize(self, counts_host):
        """counts_host: python ints per image (already clamped by the caller's sync)."""
        out = self.prepare()
        self._prepared = None
        out["nms"] = self.det.nms
        cap = self.det.cap
        ns = [min(int(c), cap) for c in counts_host]
        self._ns = ns  # what the lists below were cut to (matchers check the lists against it)
        if all(v == cap for v in ns):
            # common case (every image filled its top-k quota): one unbind instead of B slicing calls,
            # normally already done by prepare() while the device was busy
            spec = self._full_lists
            if spec is not None and spec[0] is self.det:
                out["sparse_descriptors"], out["sparse_positions"] = spec[1], spec[2]
            else:
                out["sparse_descriptors"] = list(self.sparse_desc.unbind(0))
                out["sparse_positions"] = list(self.det.positions.unbind(0))
        else:
            out["sparse_descriptors"] = [self.sparse_desc[b, :ns[b]] for b in range(self.B)]
            out["sparse_positions"] = [self.det.positions[b, :ns[b]] for b in range(self.B)]
        out._batched = self
        return out


def _mask_u8(mask, H, W):
    if mask is None:
        return None
    if mask.device.type != "cuda":
        raise RuntimeError(f"einx: the mask must live on a HIP device, got {mask.device}")
    if tuple(mask.shape[-2:]) != (H, W):
        raise ValueError(f"mask spatial size {tuple(mask.shape[-2:])} does not match the image ({H},{W})")
    if mask.dtype == torch.bool:
        return mask.contiguous().view(torch.uint8)
    # other dtypes: "non-zero is visible" -- the reference averages `score_mask.float()` over 3x3 and tests `> 0`
    # (EventExtractors.py:529-535), so a fractional 0.5 counts; a plain cast to uint8 would truncate it to 0
    return mask.ne(0).contiguous().view(torch.uint8)


class ExtractorEngine:
    """Holds the kernel-native layer images of one network and runs the batched forward."""

    use_handle = True  # one einx_extract call per network (tests may set it to False: layer by layer through the op-level ABI)

    def __init__(self, kind, *, top_k, radius, border, det_thr, ordering, cell):
        self.kind = kind
        self.cell = cell
        self.top_k, self.radius, self.border, self.det_thr, self.ordering = top_k, radius, border, det_thr, ordering
        self.backbone = []  # ConvLayer list
        self.det_head = []
        self.desc_head = []
        # wide NMS passes enqueued per forward (EINX_NMS_PASSES: tuning).  Measured: 4 instead of 8 at radius 4 leaves real work to
        # nms4_finish_kernel (one workgroup per image walking 54 tiles per pass): B=1 0.84 -> 1.01 ms, B=32 3440 -> 3347 pairs/s
        self.nms_base = max(1, int(os.environ.get("EINX_NMS_PASSES", "8")))  # 0 or less would disable the retry growth (0 * 4 = 0)
        self.nms_iters = self.nms_base
        self._handle = None
        self._handle_key = None
        self._shapes = {}
        self.valid_crop = 0  # 9: padding=0 networks (see run)

    # ------------------------------------------------------------------ handle-level C ABI (one call per forward)
    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            try:
                N.lib().einx_extractor_destroy(h)
            except Exception:
                pass

    def handle(self, scale, dilate_mask, input_div):
        """einx_extractor for this network (include/einx.h): rebuilt when a call-time parameter changes."""
        key = (float(scale), bool(dilate_mask), float(input_div), self.top_k, self.radius, self.border, self.det_thr, self.ordering)
        if self._handle is not None and self._handle_key == key:
            return self._handle
        L = N.lib()
        if self._handle:
            L.einx_extractor_destroy(self._handle)
            self._handle = None
        arr = lambda layers: (_lib.ConvDesc * len(layers))(*[l.desc for l in layers])  # noqa: E731
        bb, det, desc = arr(self.backbone), arr(self.det_head), arr(self.desc_head)
        mh = getattr(self, "merged_head0", None)
        d = _lib.ExtractorDesc(ctypes.sizeof(_lib.ExtractorDesc), self.cell, len(self.backbone), len(self.det_head), len(self.desc_head), bb, det, desc, int(bool(dilate_mask)),
                               int(self.border), int(self.radius), int(self.top_k or 0), float(self.det_thr), int(self.ordering == "xy"),
                               float(scale), float(input_div), ctypes.pointer(mh.desc) if mh is not None else None)
        h = L.einx_extractor_create(ctypes.byref(d))
        if not h:
            raise _lib.EinxError("einx_extractor_create failed: " + L.einx_last_error().decode(errors="replace"))
        self._handle, self._handle_key, self._shapes = h, key, {}
        return h

    def shapes(self, h, H, W):
        s = self._shapes.get((H, W))
        if s is None:
            s = _lib.ExtractShapes()
            _lib.check(N.lib().einx_extract_shapes(h, H, W, ctypes.byref(s)), "einx_extract_shapes")
            self._shapes[(H, W)] = s
        return s

    def _run_handle(self, h, sh, x, mask, pads, scale, dense, nms_iters, defer_dense=False):
        L = N.lib()
        dev = x.device
        B, _, H, W = x.shape
        F32 = torch.float32
        cap, hc, wc, Hp, Wp, D = sh.cap, sh.hc, sh.wc, sh.Hp, sh.Wp, sh.desc_dim
        e = lambda *shape, dt=F32: torch.empty(shape, dtype=dt, device=dev)  # noqa: E731
        bf = BatchedFeats()
        bf.kind, bf.cell, bf.B = self.kind, self.cell, B
        bf.image_size, bf.pads, bf.padded = (H, W), pads, (Hp, Wp)
        bf.scale, bf.ordering, bf.dense_lazy = float(scale), self.ordering, (isinstance(dense, str) and dense == "lazy")
        dense = bf.dense = bool(dense) and not bf.dense_lazy
        bf.feats, bf.logits, bf.raw = e(B, sh.feat_channels, hc, wc), e(B, sh.det_channels, hc, wc), e(B, D, hc, wc)
        bf.prob, bf.score = e(B, sh.det_channels, hc, wc), e(B, 1, Hp, Wp)
        if self.cell == 8:
            bf.coarse, bf.raw_cl = e(B, D, hc, wc), e(B, hc * wc, D)
        det = N.Detection()
        det.positions, det.indices = e(B, cap, 3), e(B, cap, dt=torch.int32)
        det.counts, det.thr, det.not_converged = e(B, dt=torch.int32), e(B), e(B, dt=torch.int32)
        det.nms = e(B, H, W)
        det.cap, det.padded, det.pads = cap, (Hp, Wp), pads
        bf.det = det
        bf.sparse_desc = e(B, cap, D)
        nws = int(L.einx_extract_ws_bytes(h, B, H, W, cap, int(nms_iters)))
        ws = e(nws, dt=torch.uint8)
        m8 = _mask_u8(mask, H, W)
        P = N._ptr
        bf.score_crop = e(B, 1, H, W)  # the dict's un-padded `score`, written by the score kernel (no crop + clone launch afterwards)
        out = _lib.ExtractOut(P(bf.feats), P(bf.logits), P(bf.raw), P(bf.prob), P(bf.score), P(bf.coarse), P(bf.raw_cl), P(det.nms),
                              P(det.positions), P(det.indices), P(det.counts), P(det.thr), P(det.not_converged), P(bf.sparse_desc), cap,
                              P(bf.score_crop))
        wt = getattr(self, "watch", None)
        if wt is not None and wt.n:  # `.data` edits of the weights: compared inside the call, reported through the watch's own `stale` word
            ww = _lib.WeightWatch(ctypes.sizeof(_lib.WeightWatch), wt.n, P(wt.table), P(wt.ref), P(wt.scratch), P(wt.stale))
            _lib.check(L.einx_extract_watch(h, P(x), P(m8), B, H, W, int(nms_iters), P(ws), nws, ctypes.byref(out), ctypes.byref(ww), N._stream(x)),
                       "einx_extract_watch")
            det.stale = wt.stale
        else:
            _lib.check(L.einx_extract(h, P(x), P(m8), B, H, W, int(nms_iters), P(ws), nws, ctypes.byref(out), N._stream(x)), "einx_extract")
        if dense and not defer_dense:
            bf.run_dense()
        return bf

    def redetect(self, bf, nms_iters=None):
        """NMS fix-point + top-k + compaction + sparse descriptor sampling on bf.score (also the
        retry path when the fix-point needed more passes than were enqueued: only this tail of the
        pipeline is redone, never the convolutions)."""
        det = N.detect(bf.score, top_k=self.top_k, radius=self.radius, det_thr=self.det_thr, pads=bf.pads, ordering=self.ordering,
                       nms_iters=nms_iters or self.nms_iters)
        if det.cap > 8192:
            # unbounded-capacity configuration (no top-k or detection_threshold < 1): size the
            # descriptor buffer from the real counts (one extra sync, never on the default path)
            bf.sized_from_counts = True  # (a host round trip: EIM.forward_graph refuses to capture this configuration)
            cmax = max(int(det.counts.max().item()), 1)
            det.positions = det.positions[:, :cmax].contiguous()
            det.indices = det.indices[:, :cmax].contiguous()
            det.cap = cmax
        if self.valid_crop:  # mapping_positions: both coordinates + 9 (the third column is the score); indices stay map-relative
            det.positions[:, :, :2] += float(self.valid_crop)
        bf.det = det
        bf.sparse_desc = N.desc_sample(bf.raw, det.indices, det.counts, bf.padded, bilinear=(self.cell == 8), scale=bf.scale,
                                       raw_cl=bf.raw_cl)
        return bf

    def grow_nms_iters(self):
        """The fix-point needed more passes than enqueued: quadruple the budget and keep it
        (skipped passes cost ~4 us each, a redo costs a host round trip)."""
        self.nms_iters = min(self.nms_iters * 4, 65536)
        self._calm = 0
        return self.nms_iters

    def note_converged(self):
        """A grown budget decays again: after 64 consecutive forwards that converged, halve it (never below the base),
        so one pathological image does not make every later call enqueue thousands of (skipped) passes."""
        if self.nms_iters > self.nms_base:
            self._calm = getattr(self, "_calm", 0) + 1
            if self._calm >= 64:
                self.nms_iters = max(self.nms_base, self.nms_iters // 2)
                self._calm = 0

    def run(self, x, mask, *, scale, dilate_mask, dense=False, nms_iters=None, input_div=0.0, defer_dense=False):
        """One einx_extract call (handle-level ABI) enqueues the whole network; the op-by-op path below it serves the
        unbounded-capacity configurations (no top-k: the descriptor buffer is sized from the real counts)."""
        if x.dim() != 4:
            raise ValueError(f"Expected 4D tensor, got {x.dim()}D tensor instead.")
        if x.dtype != torch.float32:
            raise TypeError("einx extractors compute in fp32; pass a float32 tensor")
        if x.device.type != "cuda":
            raise RuntimeError("einx: input must be on a HIP device (no CPU path)")
        if input_div and not x.is_contiguous():
            raise RuntimeError("einx: the input must be contiguous (it is scaled in place like the reference does)")
        x = x.contiguous()
        B, C, H, W = x.shape
        if C != self.backbone[0].cin:
            raise ValueError(f"conv expects {self.backbone[0].cin} input channels, got {C}")
        pads = N.padder_pads(H, W, self.cell)
        w0, w1, h0, h1 = pads
        Hp, Wp = H + h0 + h1, W + w0 + w1
        crop = self.valid_crop
        if crop:
            # padding=0 (EventExtractors.py:319-329, silk_extractor.py:142-152): nine un-padded 3x3 layers.  An un-padded layer
            # equals the padded one away from the border, so the padded network's maps cropped by 8 (backbone_feats) / 9
            # (logits, raw_descriptors) ARE the un-padded network's maps, value for value; everything downstream runs on the
            # (H-18) x (W-18) maps and the keypoints are shifted back by +9 (mapping_positions).
            if mask is not None:
                raise RuntimeError(f"The shape of the mask {list(mask.shape)} at index 0 does not match the shape of the indexed tensor "
                                   f"[{B}, 1, {H - 2 * crop}, {W - 2 * crop}] at index 0")  # what `score[~score_mask] = 0` raises in the reference
            if H <= 2 * crop + 8 or W <= 2 * crop + 8:
                raise ValueError("image too small for nine un-padded 3x3 convolutions")
        else:
            h = self.handle(scale, dilate_mask, input_div)
            sh = self.shapes(h, H, W)
            if sh.cap <= 8192 and self.use_handle:
                return self._run_handle(h, sh, x, mask, pads, scale, dense, nms_iters or self.nms_iters, defer_dense=defer_dense)
        if input_div:
            N.div_inplace(x, input_div)
        t = x
        first = True
        for layer in self.backbone:
            t = layer(t, fold=(h0, w0, Hp, Wp) if first else None)
            first = False
        feats = t
        d = feats
        for layer in self.det_head:
            d = layer(d)
        logits = d
        d = feats
        for layer in self.desc_head:
            d = layer(d)
        raw = d
        if crop:
            feats = feats[:, :, crop - 1:-(crop - 1), crop - 1:-(crop - 1)].contiguous()
            logits = logits[:, :, crop:-crop, crop:-crop].contiguous()
            raw = raw[:, :, crop:-crop, crop:-crop].contiguous()
            Hp, Wp = Hp - 2 * crop, Wp - 2 * crop
        bf = BatchedFeats()
        bf.kind, bf.cell, bf.B = self.kind, self.cell, B
        bf.image_size, bf.pads, bf.padded = (H, W), pads, (Hp, Wp)
        bf.shift = float(crop)
        bf.scale, bf.ordering, bf.dense_lazy = float(scale), self.ordering, (isinstance(dense, str) and dense == "lazy")
        dense = bf.dense = bool(dense) and not bf.dense_lazy
        # dense by-products first: they depend only on `raw`, so they run under the other stream's
        # convolutions instead of lengthening the latency-bound detection tail at the end of the step
        if self.cell == 8:
            bf.coarse, bf.raw_cl = N.normalize_map(raw, scale, want_cl=True)  # + channels-last raw copy for the sparse sampler
        prob, score = N.score_map(logits, mask, pads, dilate=dilate_mask, border=self.border)
        bf.feats, bf.logits, bf.raw, bf.prob, bf.score = feats, logits, raw, prob, score
        self.redetect(bf, nms_iters)
        wt = getattr(self, "watch", None)
        if wt is not None and wt.n:  # op-level path (rare configurations): the weight watch as its own launch, same word
            bf.det.stale = wt.check()
        if dense and not defer_dense:
            bf.run_dense()
        return bf
