"""numpy front-end of the CPU ORACLE (oracle/einx_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package.  Composes the C primitives into the same
pipelines the reference composes (citations = reference file:line):

  extractor_forward  ~ VGGExtractor.forward            core/modules/event_extractors/EventExtractors.py:517-624
                       VGGExtractorNP.forward          core/modules/event_extractors/EventExtractors.py:331-434
                       SuperPointv1.forward            core/modules/image_extractors/superpoint_extractor.py:345-480
                       SiLKModel.forward               core/modules/image_extractors/silk_extractor.py:177-257
  mnn                ~ NearestNeighborMatcher.forward  core/modules/matchers/MNN.py:43-140
  lightglue          ~ LightGlue.forward               core/modules/matchers/lightglue.py:522-716

Parity status: pinned by tests/golden/*.npz (captured from the reference, torch 2.10 CPU).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_f = ctypes.POINTER(ctypes.c_float)
c_i32 = ctypes.POINTER(ctypes.c_int32)
c_i64 = ctypes.POINTER(ctypes.c_int64)
c_u8 = ctypes.POINTER(ctypes.c_uint8)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libeinx_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.orc_fast_nms.restype = ctypes.c_int
    return _LIB


def _f(a):
    return None if a is None else a.ctypes.data_as(c_f)


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


# ------------------------------------------------------------------------------ primitives
def padder_pads(h, w, p):
    """Padder.__init__ (core/modules/utils/util.py:6-15) -> (w0, w1, h0, h1)."""
    hp = (((h // p) + 1) * p - h) % p
    wp = (((w // p) + 1) * p - w) % p
    return (wp // 2, wp - wp // 2, hp // 2, hp - hp // 2)


def pad_replicate(x, pads):
    x = _c(x)
    B, C, H, W = x.shape
    w0, w1, h0, h1 = pads
    out = np.empty((B, C, H + h0 + h1, W + w0 + w1), np.float32)
    lib().orc_pad_replicate(_f(x), B * C, H, W, w0, w1, h0, h1, _f(out))
    return out


def bn_fold(g, b, mean, var, eps=1e-5):
    """BatchNorm2d(eval) as per-channel affine; same fp32 op sequence as the product's repack."""
    scale = (g / np.sqrt(var + np.float32(eps))).astype(np.float32)
    shift = (b - mean * scale).astype(np.float32)
    return scale, shift


def conv_block(x, w, bias, scale=None, shift=None, relu=True, pool=False):
    x, w = _c(x), _c(w)
    B, Cin, H, W = x.shape
    Cout, _, ks, _ = w.shape
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    out = np.empty((B, Cout, Ho, Wo), np.float32)
    bias = None if bias is None else _c(bias)
    scale = None if scale is None else _c(scale)
    shift = None if shift is None else _c(shift)
    lib().orc_conv_block(_f(x), B, Cin, H, W, _f(w), _f(bias), _f(scale), _f(shift), Cout, ks, int(relu), int(pool), _f(out))
    return out


def logits_to_score(logits):
    logits = _c(logits)
    B, C, hc, wc = logits.shape
    prob = np.empty_like(logits)
    cell = 8 if C == 65 else 1
    score = np.empty((B, 1, hc * cell, wc * cell), np.float32)
    lib().orc_logits_to_score(_f(logits), B, C, hc, wc, _f(prob), _f(score))
    return prob, score


def mask_border(score, mask, pads, dilate, border):
    """In place on score [B,1,Hp,Wp]; mask [B,1,H,W] bool or None."""
    B, _, Hp, Wp = score.shape
    w0, w1, h0, h1 = pads
    H, W = Hp - h0 - h1, Wp - w0 - w1
    m = None if mask is None else _c(mask, np.uint8)
    lib().orc_mask_border(_f(score), B, Hp, Wp, None if m is None else m.ctypes.data_as(c_u8), H, W, h0, w0, int(dilate), int(border))
    return score


def fast_nms(m, radius):
    """In place on [B,H,W]; returns number of suppression iterations."""
    B, H, W = m.shape
    return lib().orc_fast_nms(_f(m), B, H, W, int(radius))


def topk_threshold(m, top_k, det_thr):
    B = m.shape[0]
    N = int(np.prod(m.shape[1:]))
    thr = np.empty((B,), np.float32)
    lib().orc_topk_threshold(_f(m), B, N, int(top_k or 0), ctypes.c_float(det_thr), _f(thr))
    return thr


def topk_capacity(N, top_k, det_thr=1.0):
    """Upper bound on survivors of `map > thr` per image (kernel K5): with the
    top-k rule alone at most N-1-lo values exceed sorted[lo]; a detection threshold below 1.0
    can lower thr further, so nothing better than N is guaranteed then."""
    if not top_k or top_k >= N or det_thr < 1.0:
        return N
    lo = ctypes.c_int()
    hi = ctypes.c_int()
    lib().orc_topk_ranks(N, int(top_k), ctypes.byref(lo), ctypes.byref(hi))
    return N - 1 - lo.value


def positions(m, pads, ordering, cap):
    B, Hp, Wp = m.shape
    w0, w1, h0, h1 = pads
    H, W = Hp - h0 - h1, Wp - w0 - w1
    pos = np.zeros((B, cap, 3), np.float32)
    idx = np.zeros((B, cap), np.int32)
    cnt = np.zeros((B,), np.int32)
    lib().orc_positions(_f(m), B, Hp, Wp, h0, w0, H, W, int(ordering == "xy"), cap, _f(pos), idx.ctypes.data_as(c_i32),
                        cnt.ctypes.data_as(c_i32))
    return pos, idx, cnt


def detect_post(score, top_k, radius, border, det_thr, pads=(0, 0, 0, 0), ordering="yx"):
    """prob_map_to_points_map + prob_map_to_positions_with_prob + unpad/filter.
    score [B,1,Hp,Wp] is modified in place by the border removal (reference quirk A10).
    Returns (nms_map [B,Hp,Wp], positions list, flat idx list, thr, iterations)."""
    B, _, Hp, Wp = score.shape
    mask_border(score, None, pads, False, border)
    m = score[:, 0].copy()
    iters = fast_nms(m, radius)
    thr = topk_threshold(m, top_k, det_thr)
    cap = topk_capacity(Hp * Wp, top_k, det_thr)
    pos, idx, cnt = positions(m, pads, ordering, cap)
    assert (cnt <= cap).all(), (cnt, cap)
    return m, [pos[b, :cnt[b]] for b in range(B)], [idx[b, :cnt[b]] for b in range(B)], thr, iters


def _pack_idx(idx_list, cap):
    B = len(idx_list)
    idx = np.zeros((B, max(cap, 1)), np.int32)
    cnt = np.zeros((B,), np.int32)
    for b, v in enumerate(idx_list):
        idx[b, :len(v)] = v
        cnt[b] = len(v)
    return idx, cnt


def desc_sample_bilinear(raw, idx_list, padded_size, scale):
    raw = _c(raw)
    B, D, hc, wc = raw.shape
    cap = max([len(v) for v in idx_list] + [1])
    idx, cnt = _pack_idx(idx_list, cap)
    out = np.zeros((B, cap, D), np.float32)
    lib().orc_desc_sample_bilinear(_f(raw), B, D, hc, wc, int(padded_size[0]), int(padded_size[1]), idx.ctypes.data_as(c_i32),
                                   cnt.ctypes.data_as(c_i32), cap, ctypes.c_float(scale), _f(out))
    return [out[b, :cnt[b]] for b in range(B)]


def desc_gather(raw, idx_list, scale):
    raw = _c(raw)
    B, D, H, W = raw.shape
    cap = max([len(v) for v in idx_list] + [1])
    idx, cnt = _pack_idx(idx_list, cap)
    out = np.zeros((B, cap, D), np.float32)
    lib().orc_desc_gather(_f(raw), B, D, H, W, idx.ctypes.data_as(c_i32), cnt.ctypes.data_as(c_i32), cap, ctypes.c_float(scale), _f(out))
    return [out[b, :cnt[b]] for b in range(B)]


def normalize_map(raw, scale):
    raw = _c(raw)
    B, D = raw.shape[:2]
    P = int(np.prod(raw.shape[2:]))
    out = np.empty_like(raw)
    lib().orc_normalize_map(_f(raw), B, D, P, ctypes.c_float(scale), _f(out))
    return out


def upsample_normalize(raw, size, scale):
    raw = _c(raw)
    B, D, hc, wc = raw.shape
    out = np.empty((B, D, size[0], size[1]), np.float32)
    lib().orc_upsample_normalize(_f(raw), B, D, hc, wc, int(size[0]), int(size[1]), ctypes.c_float(scale), _f(out))
    return out


def mnn(d0, d1, want_la=True, want_sim=False):
    d0, d1 = _c(d0), _c(d1)
    n, D = d0.shape
    m = d1.shape[0]
    m0 = np.empty((n,), np.int64)
    m1 = np.empty((m,), np.int64)
    s0 = np.empty((n,), np.float32)
    s1 = np.empty((m,), np.float32)
    la = np.empty((n + 1, m + 1), np.float32) if want_la else None
    sim = np.empty((n, m), np.float32) if want_sim else None
    lib().orc_mnn(_f(d0), n, _f(d1), m, D, m0.ctypes.data_as(c_i64), m1.ctypes.data_as(c_i64), _f(s0), _f(s1), _f(la), _f(sim))
    return dict(matches0=m0, matches1=m1, matching_scores0=s0, matching_scores1=s1, log_assignment=la, similarity=sim)


def mnn_thresh(d0, d1, ratio_thresh=None, distance_thresh=None):
    """find_nn with ratio / distance thresholds + mutual check (MNN.py:11-32); the Python scalars are squared
    in double and rounded to fp32 once, as torch does for `tensor <= scalar` / `scalar * tensor`."""
    d0, d1 = _c(d0), _c(d1)
    n, D = d0.shape
    m = d1.shape[0]
    m0, m1 = np.empty((n,), np.int64), np.empty((m,), np.int64)
    s0, s1 = np.empty((n,), np.float32), np.empty((m,), np.float32)
    r2 = np.float32(float(ratio_thresh) ** 2) if ratio_thresh else np.float32(0)
    t2 = np.float32(float(distance_thresh) ** 2) if distance_thresh else np.float32(0)
    L = lib()
    L.orc_mnn_thresh.restype = ctypes.c_int
    rc = L.orc_mnn_thresh(_f(d0), n, _f(d1), m, D, int(bool(ratio_thresh)), ctypes.c_float(r2), int(bool(distance_thresh)), ctypes.c_float(t2),
                          m0.ctypes.data_as(c_i64), m1.ctypes.data_as(c_i64), _f(s0), _f(s1))
    if rc != 0:
        raise RuntimeError("selected index k out of range")  # what torch.topk(2) raises on a single candidate
    return dict(matches0=m0, matches1=m1, matching_scores0=s0, matching_scores1=s1)


# ------------------------------------------------------------------------------ extractors
def _blk(sd, prefix, conv, bn):
    w, b = sd[f"{prefix}{conv}.weight"], sd[f"{prefix}{conv}.bias"]
    if bn is None:
        return w, b, None, None
    s, t = bn_fold(sd[f"{prefix}{bn}.weight"], sd[f"{prefix}{bn}.bias"], sd[f"{prefix}{bn}.running_mean"], sd[f"{prefix}{bn}.running_var"])
    return w, b, s, t


def vgg_net(sd, x, pool, prefix=""):
    """VGGBackBone + heads (core/modules/net/backbone.py:105-128, detector_head.py:42-48,
    descriptor_head.py:40-43); block = conv -> relu -> bn (net/vgg.py:34-38)."""
    for li in range(1, 5):
        for j in range(2):
            w, b, s, t = _blk(sd, f"{prefix}backbone.l{li}.{j}.", "0", "2")
            x = conv_block(x, w, b, s, t, relu=True, pool=(pool and j == 1 and li < 4))
    feats = x
    w, b, s, t = _blk(sd, f"{prefix}detector_head._detH1.", "0", "2")
    d = conv_block(feats, w, b, s, t, relu=True)
    w, b, s, t = _blk(sd, f"{prefix}detector_head._detH2.", "0", "1")
    logits = conv_block(d, w, b, s, t, relu=False)
    w, b, s, t = _blk(sd, f"{prefix}descriptor_head._desH1.", "0", "2")
    d = conv_block(feats, w, b, s, t, relu=True)
    w, b, s, t = _blk(sd, f"{prefix}descriptor_head._desH2.", "0", "1")
    raw = conv_block(d, w, b, s, t, relu=False)
    return feats, logits, raw


def superpoint_net(sd, x, prefix=""):
    """SuperPointv1 encoder + heads (image_extractors/superpoint_extractor.py:388-406)."""
    for name, pool in (("1a", 0), ("1b", 1), ("2a", 0), ("2b", 1), ("3a", 0), ("3b", 1), ("4a", 0), ("4b", 0)):
        x = conv_block(x, sd[f"{prefix}conv{name}.weight"], sd[f"{prefix}conv{name}.bias"], relu=True, pool=bool(pool))
    feats = x
    cpa = conv_block(feats, sd[f"{prefix}convPa.weight"], sd[f"{prefix}convPa.bias"], relu=True)
    logits = conv_block(cpa, sd[f"{prefix}convPb.weight"], sd[f"{prefix}convPb.bias"], relu=False)
    cda = conv_block(feats, sd[f"{prefix}convDa.weight"], sd[f"{prefix}convDa.bias"], relu=True)
    raw = conv_block(cda, sd[f"{prefix}convDb.weight"], sd[f"{prefix}convDb.bias"], relu=False)
    return feats, logits, raw


def silk_net(sd, x, prefix="model."):
    """SiLKVGG: ParametricVGG(no pooling, BN) + heads (silk/backbones/superpoint/vgg.py:284-290,
    magicpoint.py:95-101, superpoint.py:60-64)."""
    for i in range(4):
        for j in range(2):
            w, b, s, t = _blk(sd, f"{prefix}backbone._backbone.layers.{i}.{j}.", "0", "2")
            x = conv_block(x, w, b, s, t, relu=True)
    feats = x
    hp = f"{prefix}backbone._heads._mods."
    w, b, s, t = _blk(sd, f"{hp}logits._detH1.", "0", "2")
    d = conv_block(feats, w, b, s, t, relu=True)
    w, b, s, t = _blk(sd, f"{hp}logits._detH2.", "0", "1")
    logits = conv_block(d, w, b, s, t, relu=False)
    w, b, s, t = _blk(sd, f"{hp}raw_descriptors._desH1.", "0", "2")
    d = conv_block(feats, w, b, s, t, relu=True)
    w, b, s, t = _blk(sd, f"{hp}raw_descriptors._desH2.", "0", "1")
    raw = conv_block(d, w, b, s, t, relu=False)
    return feats, logits, raw


def extractor_forward(kind, sd, x, mask, *, top_k, radius=4, border=4, det_thr=1.0, ordering="yx", scale=1.0, dense=False, padding=1):
    """kind in {'vgg','vgg_np','superpointv1','silk'}.  x [B,C,H,W] fp32 (image: 0..255; modified
    in place for 'superpointv1' like the reference, superpoint_extractor.py:372).  Returns the
    reference's output dict (numpy), positions as lists.
    padding=0 (cell-1 nets only: nine un-padded 3x3 convolutions, EventExtractors.py:319-329 / silk_extractor.py:142-152):
    an un-padded 3x3 layer equals the padded one away from the border, so the interior of the padded network's maps
    (8 pixels in for backbone_feats, 9 for logits / raw_descriptors) IS the un-padded network, value for value; the
    detector runs on the (H-18)x(W-18) maps and the keypoints are shifted by +9 (mapping_positions)."""
    B, _, H, W = x.shape
    cell = 8 if kind in ("vgg", "superpointv1") else 1
    if kind == "silk":
        mask = None  # SiLKModel.forward(self, image, *args, **kwargs) never looks at the mask it is handed (silk_extractor.py:177)
    if kind == "superpointv1":
        np.divide(x, np.float32(255.0), out=x)  # in place, through the array's strides (superpoint_extractor.py:372)
        if x.shape[1] == 3:
            # rgb_to_grayscale (:375-376; kornia 0.7.1, kornia/color/gray.py: `w_r * r + w_g * g + w_b * b` with the fp32
            # weights (0.299, 0.587, 0.114)): three fp32 products, summed left to right, every operation rounded to fp32
            r, g, b_ = x[:, 0:1], x[:, 1:2], x[:, 2:3]
            x = (np.float32(0.299) * r + np.float32(0.587) * g) + np.float32(0.114) * b_
        x = np.ascontiguousarray(x)
    elif kind == "silk":
        x = x / np.float32(255.0)
    pads = padder_pads(H, W, cell)
    xp = pad_replicate(x, pads)
    Hp, Wp = xp.shape[-2:]
    if kind == "vgg":
        feats, logits, raw = vgg_net(sd, xp, pool=True)
    elif kind == "vgg_np":
        feats, logits, raw = vgg_net(sd, xp, pool=False)
    elif kind == "superpointv1":
        feats, logits, raw = superpoint_net(sd, xp)
    elif kind == "silk":
        feats, logits, raw = silk_net(sd, xp)
    else:
        raise ValueError(kind)
    if padding == 0:
        if cell != 1:
            raise NotImplementedError("padding=0 with pooling")
        if mask is not None:
            raise RuntimeError("The shape of the mask [%d, 1, %d, %d] at index 0 does not match the shape of the indexed tensor" % (B, H, W))
        feats = np.ascontiguousarray(feats[:, :, 8:-8, 8:-8])
        logits = np.ascontiguousarray(logits[:, :, 9:-9, 9:-9])
        raw = np.ascontiguousarray(raw[:, :, 9:-9, 9:-9])
        Hp, Wp = Hp - 18, Wp - 18
    prob, score = logits_to_score(logits)
    if mask is not None:
        mask_border(score, mask, pads, dilate=kind in ("vgg", "vgg_np"), border=0)
    nms, pos, idx, thr, iters = detect_post(score, top_k, radius, border, det_thr, pads, ordering)
    if cell == 8:
        sparse = desc_sample_bilinear(raw, idx, (Hp, Wp), scale)
    else:
        sparse = desc_gather(raw, idx, scale)
    w0, w1, h0, h1 = pads
    if padding == 0:
        for p_ in pos:
            p_[:, 0] += np.float32(9.0)
            p_[:, 1] += np.float32(9.0)
    out = {
        "image_size": [np.array([H, W], np.int64)] * B,
        "backbone_feats": feats, "logits": logits, "raw_descriptors": raw,
        "probability": score if cell == 1 else prob,
        "score": score[:, :, h0:Hp - h1, w0:Wp - w1].copy(),
        "nms": nms[:, h0:Hp - h1, w0:Wp - w1].copy(),
        "sparse_descriptors": sparse, "sparse_positions": pos,
        "_thr": thr, "_nms_iters": iters, "_idx": idx,
    }
    if cell == 8:
        out["coarse_descriptors"] = normalize_map(raw, scale)
    if dense:
        if cell == 8:
            nd = upsample_normalize(raw, (Hp, Wp), scale)
        else:
            nd = normalize_map(raw, scale)
        out["normalized_descriptors"] = nd[:, :, h0:Hp - h1, w0:Wp - w1].copy()
    return out


# ------------------------------------------------------------------------------ LightGlue
def lightglue(sd, kpts0, desc0, kpts1, desc1, size0=(260, 346), size1=(260, 346), n_layers=9, heads=4, filter_threshold=0.0,
              prefix="", capture_layers=()):
    """LightGlue.forward for one pair (B=1).  kpts [n,>=2] (first two columns used), desc [n,Din]."""
    L = lib()
    k0, k1 = _c(kpts0), _c(kpts1)
    x0, x1 = _c(desc0).copy(), _c(desc1).copy()
    n, m = x0.shape[0], x1.shape[0]
    g = lambda k: _c(sd[prefix + k])  # noqa: E731
    if (prefix + "input_proj.weight") in sd:
        w, b = g("input_proj.weight"), g("input_proj.bias")
        y0 = np.empty((n, w.shape[0]), np.float32)
        y1 = np.empty((m, w.shape[0]), np.float32)
        L.orc_linear(_f(x0), n, x0.shape[1], _f(w), _f(b), w.shape[0], _f(y0))
        L.orc_linear(_f(x1), m, x1.shape[1], _f(w), _f(b), w.shape[0], _f(y1))
        x0, x1 = y0, y1
    d = x0.shape[1]
    Wr = g("posenc.Wr.weight")
    dh = d // heads  # lightglue.py:456
    assert Wr.shape == (dh // 2, 2) and d == heads * dh
    enc0 = np.empty((2, n, dh), np.float32)
    enc1 = np.empty((2, m, dh), np.float32)
    L.orc_lg_posenc_dh(_f(k0), k0.shape[1], n, ctypes.c_float(size0[0]), ctypes.c_float(size0[1]), _f(Wr), dh, _f(enc0))
    L.orc_lg_posenc_dh(_f(k1), k1.shape[1], m, ctypes.c_float(size1[0]), ctypes.c_float(size1[1]), _f(Wr), dh, _f(enc1))
    captured = {}
    for i in range(n_layers):
        p = f"transformers.{i}.self_attn."
        sa = [g(p + s) for s in ("Wqkv.weight", "Wqkv.bias", "out_proj.weight", "out_proj.bias", "ffn.0.weight", "ffn.0.bias",
                                 "ffn.1.weight", "ffn.1.bias", "ffn.3.weight", "ffn.3.bias")]
        L.orc_lg_self_block(_f(x0), n, d, heads, _f(enc0), *[_f(a) for a in sa])
        L.orc_lg_self_block(_f(x1), m, d, heads, _f(enc1), *[_f(a) for a in sa])
        p = f"transformers.{i}.cross_attn."
        ca = [g(p + s) for s in ("to_qk.weight", "to_qk.bias", "to_v.weight", "to_v.bias", "to_out.weight", "to_out.bias",
                                 "ffn.0.weight", "ffn.0.bias", "ffn.1.weight", "ffn.1.bias", "ffn.3.weight", "ffn.3.bias")]
        L.orc_lg_cross_block(_f(x0), n, _f(x1), m, d, heads, *[_f(a) for a in ca])
        if i in capture_layers:
            captured[i] = (x0.copy(), x1.copy())
    p = f"log_assignment.{n_layers - 1}."
    Wp, bp, wm, bm = g(p + "final_proj.weight"), g(p + "final_proj.bias"), g(p + "matchability.weight"), g(p + "matchability.bias")
    scores = np.empty((n + 1, m + 1), np.float32)
    m0 = np.empty((n,), np.int64)
    m1 = np.empty((m,), np.int64)
    s0 = np.empty((n,), np.float32)
    s1 = np.empty((m,), np.float32)
    L.orc_lg_assign(_f(x0), n, _f(x1), m, d, _f(Wp), _f(bp), _f(wm), _f(bm), ctypes.c_float(filter_threshold), _f(scores),
                    m0.ctypes.data_as(c_i64), m1.ctypes.data_as(c_i64), _f(s0), _f(s1))
    return dict(matches0=m0, matches1=m1, matching_scores0=s0, matching_scores1=s1, log_assignment=scores,
                ref_descriptors0=x0, ref_descriptors1=x1, enc0=enc0, enc1=enc1, layers=captured)


def matched_kpts(kpts0, kpts1, matches0, cols):
    """ascending-i gather of matched keypoints (MNN.py:119-129 -> 3 columns; lightglue.py:690-698 -> 2)."""
    sel = np.nonzero(matches0 > -1)[0]
    return kpts0[sel][:, :cols], kpts1[matches0[sel]][:, :cols]


# ------------------------------------------------------------------------------ un-frozen Matcher branch
def normalize_rows(x, scale):
    x = _c(x)
    out = np.empty_like(x)
    if x.shape[0]:
        lib().orc_normalize_rows(_f(x), x.shape[0], x.shape[1], ctypes.c_float(scale), _f(out))
    return out


def normalize_keypoints(kpts, size):
    """lightglue.py:137-148: (kpt - size/2) / (max(size)/2), first two columns."""
    k = _c(kpts)[..., :2]
    sh = np.array([np.float32(size[0]) / np.float32(2), np.float32(size[1]) / np.float32(2)], np.float32)
    sc = np.float32(max(size[0], size[1])) / np.float32(2)
    return ((k - sh) / sc).astype(np.float32)


def pad_positions(pos, length, u, image_size, mode="random"):
    """Matcher.pad_sparse_positions_to_length (Matchers.py:67-98).  u: the [length-n,2] uniform draws
    the reference takes from torch.rand; padding rows = (u0*size0, u1*size1, 0)."""
    pos = _c(pos)
    n = pos.shape[0]
    if n > length:
        return pos[:length]
    if n == length:
        return pos
    r = length - n
    if mode == "zeros":
        pad = np.zeros((r, 3), np.float32)
    else:
        u = _c(u)
        pad = np.concatenate([u * np.array([image_size[0], image_size[1]], np.float32), np.zeros((r, 1), np.float32)], 1)
    return np.concatenate([pos, pad.astype(np.float32)], 0)


def pad_descriptors(desc, length, g, scale, mode="random"):
    """Matcher.pad_sparse_descriptors_to_length (Matchers.py:100-131).  g: the [length-n,C] normal
    draws (torch.randn); padding rows = F.normalize(g) * scale."""
    desc = _c(desc)
    n = desc.shape[0]
    if n > length:
        return desc[:length]
    if n == length:
        return desc
    pad = np.zeros((length - n, desc.shape[1]), np.float32) if mode == "zeros" else normalize_rows(g, scale)
    return np.concatenate([desc, pad], 0)


def mnn_stacked(kpts0, desc0, kpts1, desc1):
    """NearestNeighborMatcher.forward on stacked [B,n,*] inputs with B > 1 (MNN.py:88-140)."""
    B = desc0.shape[0]
    rs = [mnn(desc0[b], desc1[b], want_la=True, want_sim=True) for b in range(B)]
    out = {k: np.stack([r[k] for r in rs]) for k in ("matches0", "matches1", "matching_scores0", "matching_scores1",
                                                    "log_assignment", "similarity")}
    mk = [matched_kpts(kpts0[b], kpts1[b], rs[b]["matches0"], 3) for b in range(B)]
    out["matched_kpts0"] = [a for a, _ in mk]
    out["matched_kpts1"] = [b_ for _, b_ in mk]
    return out


def lightglue_stacked(sd, kpts0, desc0, kpts1, desc1, size0, size1, n_layers=9, training=True, prefix=""):
    """LightGlue.forward on stacked inputs with B > 1 (lightglue.py:522-716): matched keypoints in
    normalised coordinates (:677-687); ref_descriptors of every layer when training (:626-629)."""
    B = desc0.shape[0]
    cap = tuple(range(n_layers)) if training else (n_layers - 1,)
    rs = [lightglue(sd, kpts0[b], desc0[b], kpts1[b], desc1[b], size0=size0, size1=size1, n_layers=n_layers, prefix=prefix,
                    capture_layers=cap) for b in range(B)]
    out = {k: np.stack([r[k] for r in rs]) for k in ("matches0", "matches1", "matching_scores0", "matching_scores1", "log_assignment")}
    out["ref_descriptors0"] = np.stack([np.stack([r["layers"][i][0] for i in cap]) for r in rs])
    out["ref_descriptors1"] = np.stack([np.stack([r["layers"][i][1] for i in cap]) for r in rs])
    mk = [matched_kpts(normalize_keypoints(kpts0[b], size0), normalize_keypoints(kpts1[b], size1), rs[b]["matches0"], 2) for b in range(B)]
    out["matched_kpts0"] = [a for a, _ in mk]
    out["matched_kpts1"] = [b_ for _, b_ in mk]
    out["prune0"] = np.full(out["matching_scores0"].shape, n_layers, np.float32)
    out["prune1"] = np.full(out["matching_scores1"].shape, n_layers, np.float32)
    return out


# ------------------------------------------------------------------------------ event representation
def voxel_grid(events, input_size, normalize=True):
    """events_to_voxel_grid (datasets/representations.py:67-124); events: dict of numpy arrays."""
    bins, H, W = (int(v) for v in input_size)
    x, y = _c(events["x"]), _c(events["y"])
    t, p = _c(events["t"], np.float64), _c(events["p"])
    grid = np.empty((bins, H, W), np.float32)
    lib().orc_voxel_grid(_f(x), _f(y), t.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), _f(p), ctypes.c_longlong(len(x)), bins, H, W,
                         int(normalize), _f(grid))
    return grid


def events_mask(events, resolution):
    W, H = (int(v) for v in resolution)
    x, y = _c(events["x"]), _c(events["y"])
    mask = np.empty((H, W), np.uint8)
    lib().orc_events_mask(_f(x), _f(y), ctypes.c_longlong(len(x)), H, W, mask.ctypes.data_as(c_u8))
    return mask.astype(bool)


# ------------------------------------------------------------------------------ evaluation metrics
def pair_metrics(k0, k1, d0, d1, mk0, mk1, size0, size1, hom=None, mma_thr=(1, 3), vdd_thr=(1, 3), kp_yx=True):
    """MR, MMA@t, VDD (repeatability, distance, angle)@t for one pair (core/metrics/*)."""
    k0, k1, d0, d1 = _c(k0), _c(k1), _c(d0), _c(d1)
    mk0, mk1 = _c(mk0), _c(mk1)
    cols = mk0.shape[1] if mk0.ndim == 2 and mk0.shape[0] else 3
    mt, vt = _c(list(mma_thr)), _c(list(vdd_thr))
    out = np.zeros((1 + len(mma_thr) + 3 * len(vdd_thr),), np.float64)
    h = None if hom is None else _c(hom).reshape(9)
    lib().orc_pair_metrics(_f(k0), k0.shape[0], _f(k1), k1.shape[0], _f(d0), _f(d1), d0.shape[1], _f(mk0), _f(mk1), mk0.shape[0], cols, _f(h),
                           int(size0[0]), int(size0[1]), int(size1[0]), int(size1[1]), int(kp_yx), _f(mt), len(mma_thr), _f(vt), len(vdd_thr),
                           out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return out
