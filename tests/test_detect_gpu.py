"""GPU tests (-m gpu), component: detect.
SURVEY 8a rows A2, A7-A13, A17 (mask dilation, probabilities, pixel shuffle, border, fast_nms fix-point, top-k threshold, positions, unpad): detect.hip.
(Round 6 regrouped the per-round files test_gpu_parity / test_r2..r5_gpu by component; shared helpers live in gpu_support.py.)"""
import numpy as np
import pytest
import torch
import json
import os
import sys
import statistics
import subprocess
import time
from importlib import import_module

from helpers import (score_map, sub_dict, synth, tie_map)
from gpu_support import (DEV, POST, TIES, _Z, _assert_feats_equal_oracle, _np, _ramp, _rng, _serpentine, _t, pkg)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(POST.cases))
def test_post_golden_bit_exact(name):
    """reference-named helper API -> HIP kernels -> must equal the reference's own outputs."""
    from importlib import import_module
    du = import_module(pkg.__name__ + ".core.modules.utils.detector_util")
    c = POST.cases[name]
    score = _t(score_map(c))
    nms = du.prob_map_to_points_map(score, prob_thresh=c["thr"], nms_dist=c["radius"], border_dist=c["border"], use_fast_nms=True,
                                    top_k=(c["k"] or None))
    pos = du.prob_map_to_positions_with_prob(nms, threshold=0.0, ordering=c.get("ordering", "yx"))
    counts = POST[f"{name}.counts"]
    assert [int(p.shape[0]) for p in pos] == counts.tolist()
    got = np.concatenate([_np(p) for p in pos], 0)
    assert np.array_equal(got, POST[f"{name}.positions"])
    flat = _np(nms).reshape(-1)
    nz = np.nonzero(flat)[0]
    assert np.array_equal(nz, POST[f"{name}.nms_idx"])
    assert np.array_equal(flat[nz], POST[f"{name}.nms_val"])
    # border removal happened in place on the caller's tensor (reference side effect)
    assert abs(float(_np(score).astype(np.float64).sum()) - float(POST[f"{name}.score_sum"][0])) < 1e-6 * score.numel()


def test_detect_fused_matches_oracle(oracle):
    """fused einx_detect (NMS + threshold + compaction + unpad/filter) vs the oracle, with padding."""
    s = synth.uniform01(77, (3, 1, 72, 96)) ** 4
    pads = (3, 3, 2, 2)
    sc = s.copy()
    oracle.mask_border(sc, None, pads, False, 4)
    exp_nms, exp_pos, exp_idx, exp_thr, iters = oracle.detect_post(sc.copy(), 60, 4, 4, 1.0, pads, "yx")
    d = pkg.native.detect(_t(sc), top_k=60, radius=4, det_thr=1.0, pads=pads)
    cnt = _np(d.counts)
    assert cnt.tolist() == [len(p) for p in exp_pos]
    assert not _np(d.not_converged).any()
    for b in range(3):
        assert np.array_equal(_np(d.positions[b, :cnt[b]]), exp_pos[b])
        assert np.array_equal(_np(d.indices[b, :cnt[b]]), exp_idx[b])
    assert np.array_equal(_np(d.thr), exp_thr)
    assert np.array_equal(_np(d.nms), exp_nms[:, 2:-2, 3:-3])


def test_nms_long_suppression_chains(oracle):
    """a monotone ramp forces a long suppression chain (more passes than any enqueued budget).  Radius 4 (every shipped
    configuration): the device-side finisher completes the fix-point inside the same einx_detect call -- no flag, no host
    round trip, whatever the wide-pass budget.  Other radii: too few passes raise not_converged and the helper API
    converges by re-running with more passes."""
    from importlib import import_module
    du = import_module(pkg.__name__ + ".core.modules.utils.detector_util")
    H, W = 16, 200
    m = np.zeros((2, H, W), np.float32)
    m[0, 8, 4:196] = np.linspace(0.1, 0.9, 192, dtype=np.float32)  # strictly increasing along the row
    m[1, 3, 10] = 0.5  # a second image that converges at once (the finisher must leave it alone)
    exp = m.copy()
    it = oracle.fast_nms(exp, 4)
    assert it > 8
    for iters in (1, 2, 3, 8):
        d = pkg.native.detect(_t(m), top_k=0, radius=4, det_thr=float("-inf"), cap=1, nms_iters=iters)
        assert _np(d.not_converged).tolist() == [0, 0], iters
        assert np.array_equal(_np(d.nms), exp), iters
    exp2 = m.copy()
    it2 = oracle.fast_nms(exp2, 2)
    assert it2 > 4
    d = pkg.native.detect(_t(m), top_k=0, radius=2, det_thr=float("-inf"), cap=1, nms_iters=2)
    assert int(_np(d.not_converged)[0]) != 0 and int(_np(d.not_converged)[1]) == 0
    got = du.fast_nms(_t(m)[:, None], 2)
    assert np.array_equal(_np(got)[:, 0], exp2)
    assert np.array_equal(_np(du.fast_nms(_t(m)[:, None], 4))[:, 0], exp)


def test_nms_finisher_on_tie_heavy_full_size_maps(oracle):
    """quantised full-size maps need 14-17 passes: finished on the device, bit-equal to the oracle, for every image of a batch
    in which only some images are tie-heavy; positions / counts of the fused call agree too."""
    B, Hp, Wp = 6, 264, 352
    s = synth.uniform01(77, (B, 1, Hp, Wp))
    s[::2] = np.floor(s[::2] * np.float32(16.0)) / np.float32(16.0)  # images 0, 2, 4: 16 levels -> ties everywhere, ~25 passes
    score = s.copy()
    nms, pos, idx, thr, iters = oracle.detect_post(score, 0, 4, 4, 0.0, pads=(3, 3, 2, 2))
    assert iters > 16
    st = _t(s.copy())
    pkg.native.remove_border(st, 4)
    d = pkg.native.detect(st, top_k=0, radius=4, det_thr=0.0, pads=(3, 3, 2, 2), nms_iters=8)
    assert _np(d.not_converged).tolist() == [0] * B
    cnt = _np(d.counts)
    assert cnt.tolist() == [len(p) for p in pos]
    for b in range(B):
        assert np.array_equal(_np(d.positions)[b, :cnt[b]], pos[b])
    assert np.array_equal(_np(d.nms), nms[:, 2:Hp - 2, 3:Wp - 3])


def test_detect_generic_path_dense_and_negative(oracle):
    """dense maps (more non-zeros than the LDS candidate list holds) and negative values take the
    generic radix-select path of einx_detect; compare with the oracle bit for bit."""
    for seed, lo_val, radius, k in ((91, 0.0, 0, 700), (92, -0.5, 0, 300), (93, -0.2, 2, 50)):
        s = synth.uniform(seed, (2, 1, 120, 136), lo_val, 1.0)
        exp_nms, exp_pos, exp_idx, exp_thr, _ = oracle.detect_post(s.copy(), k, radius, 0, 1.0)
        d = pkg.native.detect(_t(s), top_k=k, radius=radius, det_thr=1.0)
        cnt = _np(d.counts)
        assert cnt.tolist() == [len(p) for p in exp_pos]
        assert np.array_equal(_np(d.thr), exp_thr)
        for b in range(2):
            assert np.array_equal(_np(d.positions[b, :cnt[b]]), exp_pos[b])
        assert np.array_equal(_np(d.nms), exp_nms)


def test_xy_ordering_threshold_and_single_image(oracle):
    """ordering='xy', an active detection_threshold (capacity = whole map) and B=1 odd-sized input."""
    cfg = pkg.default_config("SP_MNN", event_channels=5)
    for sec in (cfg.event_extractor.vgg, cfg.image_extractor.superpointv1):
        sec.ordering = "xy"
        sec.detection_threshold = 0.02
        sec.detection_top_k = 200
    model = pkg.EIM(cfg, device=DEV).eval()
    sdn = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=21)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()}, strict=False)
    for ext in (model.event_extractor.extractor, model.image_extractor.extractor):
        ext.dense_outputs = False
    ev, mask = synth.synth_events(41, 1, 5, 75, 93)
    img = synth.synth_image(41, 1, 75, 93)
    ef, imf, m = model(_t(ev), _t(img), _t(mask))
    oe = oracle.extractor_forward("vgg", sub_dict(sdn, "event_extractor.extractor."), ev.copy(), mask, top_k=200, det_thr=0.02, ordering="xy")
    oi = oracle.extractor_forward("superpointv1", sub_dict(sdn, "image_extractor.extractor."), img.copy(), None, top_k=200, det_thr=0.02,
                                  ordering="xy")
    _assert_feats_equal_oracle(ef, oe)
    _assert_feats_equal_oracle(imf, oi)
    p = _np(ef["sparse_positions"][0])
    assert p[:, 0].max() > 75  # first column is x for 'xy' ordering (W=93 > H=75)


# ------------------------------------------------------------------ r2 fixtures: tie maps with survivors
@pytest.mark.parametrize("name", list(TIES))
def test_tie_maps_with_survivors_vs_reference(name):
    from importlib import import_module
    du = import_module(pkg.__name__ + ".core.modules.utils.detector_util")
    c = TIES[name]
    score = _t(tie_map(c))
    nms = du.prob_map_to_points_map(score, prob_thresh=c["thr"], nms_dist=c["radius"], border_dist=c["border"], use_fast_nms=True,
                                    top_k=(c["k"] or None))
    pos = du.prob_map_to_positions_with_prob(nms, threshold=0.0, ordering="yx")
    counts = _Z[f"{name}.counts"]
    assert counts.sum() > 0
    assert [int(p.shape[0]) for p in pos] == counts.tolist()
    assert np.array_equal(np.concatenate([_np(p) for p in pos], 0), _Z[f"{name}.positions"])
    flat = _np(nms).reshape(-1)
    nz = np.nonzero(flat)[0]
    assert np.array_equal(nz, _Z[f"{name}.nms_idx"])
    assert np.array_equal(flat[nz], _Z[f"{name}.nms_val"])


# ------------------------------------------------------------------ score map: events mask placement / dilation, div_inplace sizes
@pytest.mark.parametrize("C,hc,wc,pads", [(65, 5, 7, (3, 3, 2, 2)), (65, 33, 44, (3, 3, 2, 2)), (65, 4, 6, (0, 0, 0, 0)), (1, 37, 53, (0, 0, 0, 0)),
                                          (65, 3, 5, (1, 7, 5, 3))], ids=["c65_5x7", "c65_33x44", "c65_nopad", "c1_37x53", "c65_lopsided_pads"])
@pytest.mark.parametrize("dilate", [False, True])
def test_score_map_mask_placements_vs_oracle(oracle, C, hc, wc, pads, dilate):
    """softmax / sigmoid + pixel shuffle + `score[~mask] = 0` (mask zero-padded, optionally dilated 3x3 inside the padded map,
    EventExtractors.py:544-562) + border: single events in every corner, on the image edges, next to the padding, isolated
    pixels and a random 2 % mask; the batched mask loads of the kernels must give exactly the per-pixel walk of the oracle."""
    N = pkg.native
    B = 3
    cell = 8 if C == 65 else 1
    Hp, Wp = hc * cell, wc * cell
    w0, w1, h0, h1 = pads
    H, W = Hp - h0 - h1, Wp - w0 - w1
    rng = np.random.default_rng(C * 1000 + hc * 10 + int(dilate))
    logits = rng.standard_normal((B, C, hc, wc)).astype(np.float32) * 2
    mask = np.zeros((B, 1, H, W), bool)
    for y, x in ((0, 0), (0, W - 1), (H - 1, 0), (H - 1, W - 1), (H // 2, 0), (0, W // 2), (H // 2, W // 2), (H - 1, W // 3), (H // 3, W - 1)):
        mask[0, 0, y, x] = True
    mask[1, 0] = rng.random((H, W)) < 0.02
    mask[2, 0, ::7, ::5] = True
    for border in (0, 4):
        prob, score = N.score_map(_t(logits), _t(mask), pads, dilate=dilate, border=border)
        eprob, escore = oracle.logits_to_score(logits)
        oracle.mask_border(escore, mask, pads, dilate, border)
        # cell-1 networks: `probability` aliases `score` in the reference (depth_to_space returns its input), zeros included
        assert np.array_equal(prob.cpu().numpy(), eprob if C == 65 else escore)
        assert np.array_equal(score.cpu().numpy(), escore), f"border {border}"
        assert float(escore.max()) > 0.0
    _, s_none = N.score_map(_t(logits), None, pads, dilate=dilate, border=0)
    assert np.array_equal(s_none.cpu().numpy(), oracle.logits_to_score(logits)[1])


# ------------------------------------------------------------------ non-default detector parameters through the whole extractor
@pytest.mark.parametrize("cfg_name,radius,border,k,thr", [("SP_MNN", 2, 8, 300, 1.0), ("SP_MNN", 0, 0, 500, 1.0), ("SP_MNN", 3, 1, 0, 0.02),
                                                           ("SiLK_MNN", 1, 6, 400, 1.0)], ids=lambda v: str(v))
def test_extractors_with_other_detector_parameters_vs_oracle(oracle, cfg_name, radius, border, k, thr):
    """nms_radius / remove_borders / detection_top_k / detection_threshold other than the shipped 4 / 4 / 1024 / 1.0 (radius 0 = no
    NMS, top_k 0 with an active threshold = unbounded capacity path) through both extractors vs the oracle."""
    from helpers import sub_dict
    cfg = pkg.default_config(cfg_name, event_channels=5)
    et, it = cfg.event_extractor.type, cfg.image_extractor.type
    for sec in (cfg.event_extractor[et], cfg.image_extractor[it]):
        sec.nms_radius, sec.remove_borders, sec.detection_top_k, sec.detection_threshold = radius, border, (k or None), thr
    model = pkg.EIM(cfg, device=DEV).eval()
    sd = synth.synth_state_dict([(kk, tuple(v.shape)) for kk, v in model.state_dict().items()], seed=radius * 10 + border)
    model.load_state_dict({kk: torch.from_numpy(v) for kk, v in sd.items()}, strict=False)
    for ext in (model.event_extractor.extractor, model.image_extractor.extractor):
        ext.dense_outputs = False
    H, W, B = 72, 104, 2
    ev, mask = synth.synth_events(55 + radius, B, 5, H, W)
    img = synth.synth_image(55 + radius, B, H, W)
    ef = model.event_extractor(_t(ev), _t(mask))
    imf = model.image_extractor(_t(img))
    kw = dict(top_k=k, radius=radius, border=border, det_thr=thr)
    oe = oracle.extractor_forward(et, sub_dict(sd, "event_extractor.extractor."), ev.copy(), mask, scale=cfg.event_extractor[et].descriptor_scale_factor, **kw)
    oi = oracle.extractor_forward(it, sub_dict(sd, "image_extractor.extractor."), img.copy(), None, scale=cfg.image_extractor[it].descriptor_scale_factor, **kw)
    for got, exp in ((ef, oe), (imf, oi)):
        assert np.array_equal(got["score"].cpu().numpy(), exp["score"])
        assert np.array_equal(got["nms"].cpu().numpy(), exp["nms"])
        for b in range(B):
            assert np.array_equal(got["sparse_positions"][b].cpu().numpy(), exp["sparse_positions"][b]), f"image {b}"
            assert np.array_equal(got["sparse_descriptors"][b].cpu().numpy(), exp["sparse_descriptors"][b])
    assert sum(len(p) for p in oe["sparse_positions"]) > 0


def test_nms_slow_converging_maps_finish_on_device_or_through_the_retry(oracle):
    """fast_nms' fix-point (detector_util.py:286-335) on maps that need hundreds of passes.
    * a monotone ramp on a 24x1600 map needs 324 passes: more than the 8 wide + 256 finisher passes of round 3, fewer than the
      finisher's bound of max(256, Hp + Wp) -> converges inside ONE einx_detect call, no host round trip;
    * a serpentine ramp on 96x96 needs 380 > 8 + 256: einx_detect reports not_converged, the callers' retry (budget x4 per
      round, detector_util.fast_nms here, EIM / NativeExtractor.forward alike) reaches the oracle's fix-point."""
    from importlib import import_module
    du = import_module(pkg.__name__ + ".core.modules.utils.detector_util")
    s = _ramp(24, 1600)
    exp_nms, _, _, _, iters = oracle.detect_post(s.copy(), 0, 4, 0, 0.0)
    assert 264 < iters < 1624
    d = pkg.native.detect(_t(s[:, 0]), top_k=0, radius=4, det_thr=float("-inf"), cap=1, nms_iters=8)
    assert int(d.not_converged.sum()) == 0
    assert np.array_equal(_np(d.nms).reshape(exp_nms.shape), exp_nms)
    s = _serpentine(96, 96)
    exp_nms, _, _, _, iters = oracle.detect_post(s.copy(), 0, 4, 0, 0.0)
    assert iters > 8 + 256
    d = pkg.native.detect(_t(s[:, 0]), top_k=0, radius=4, det_thr=float("-inf"), cap=1, nms_iters=8)
    assert int(d.not_converged.sum()) == 1  # honest: the bounded finisher gave up
    got = du.fast_nms(_t(s), nms_dist=4)
    assert np.array_equal(_np(got).reshape(exp_nms.shape), exp_nms)
    # the same map inside a batch next to an ordinary one: only the slow image is redone, both equal the oracle
    both = np.concatenate([s, _ramp(96, 96)], 0)
    exp_b, _, _, _, _ = oracle.detect_post(both.copy(), 0, 4, 0, 0.0)
    assert np.array_equal(_np(du.fast_nms(_t(both), nms_dist=4)).reshape(exp_b.shape), exp_b)


def test_detector_attributes_assigned_between_forwards_take_effect(oracle):
    """Found by tools/fuzz_parity.py: the reference's extractors read `detection_top_k`, `nms_radius`, `remove_borders`,
    `detection_threshold` and `ordering` in every forward (EventExtractors.py:545-556, superpoint_extractor.py:388-406), so assigning one
    between two forwards changes the next; the native engine was built once from the values at the first forward and kept them.
    It now follows the module's attributes at every call (the native handle is re-keyed)."""
    from helpers import sub_dict, synth
    cfg = pkg.default_config("SP_MNN", event_channels=5)
    model = pkg.EIM(cfg, device=DEV).eval()
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=47)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    ev, mask = synth.synth_events(47, 2, 5, 120, 152)
    img = synth.synth_image(47, 2, 120, 152)
    esd, isd = sub_dict(sd, "event_extractor.extractor."), sub_dict(sd, "image_extractor.extractor.")
    settings = [dict(top_k=1024, radius=4, border=4, det_thr=1.0, ordering="yx"), dict(top_k=37, radius=4, border=4, det_thr=1.0, ordering="yx"),
                dict(top_k=37, radius=2, border=9, det_thr=1.0, ordering="xy"), dict(top_k=400, radius=0, border=0, det_thr=0.02, ordering="yx"),
                dict(top_k=1024, radius=4, border=4, det_thr=1.0, ordering="yx")]
    counts = []
    for st in settings:
        for ext in (model.event_extractor.extractor, model.image_extractor.extractor):
            ext.detection_top_k, ext.nms_radius, ext.remove_borders = st["top_k"], st["radius"], st["border"]
            ext.detection_threshold, ext.ordering = st["det_thr"], st["ordering"]
        for fwd in (model.__call__, model.forward_graph):
            if st["det_thr"] < 1.0 and fwd == model.forward_graph:  # capacity = the whole map: sized from the real counts, not capturable
                with pytest.raises(NotImplementedError, match="bounded keypoint capacity"):
                    fwd(_t(ev), _t(img.copy()), _t(mask))
                continue
            ef, imf, m = fwd(_t(ev), _t(img.copy()), _t(mask))
            oe = oracle.extractor_forward("vgg", esd, ev.copy(), mask, **st)
            oi = oracle.extractor_forward("superpointv1", isd, img.copy(), None, **st)
            for got, exp in ((ef, oe), (imf, oi)):
                for b in range(2):
                    assert np.array_equal(_np(got["sparse_positions"][b]), exp["sparse_positions"][b]), st
                    assert np.array_equal(_np(got["sparse_descriptors"][b]), exp["sparse_descriptors"][b]), st
        counts.append(len(oe["sparse_positions"][0]))
    assert counts[1] <= 37 < counts[0] and counts[-1] == counts[0]


@pytest.mark.parametrize("seed", list(range(16)))
def test_random_detect(oracle, seed):
    r = _rng(2000 + seed)
    B = int(r.integers(1, 5))
    cell = int(r.choice([1, 8]))
    H, W = int(r.integers(24, 150)), int(r.integers(24, 180))
    pads = pkg.native.padder_pads(H, W, cell)
    Hp, Wp = H + pads[2] + pads[3], W + pads[0] + pads[1]
    radius = int(r.choice([0, 1, 2, 3, 4, 4, 4, 4]))
    border = int(r.integers(0, 6))
    kind = r.choice(["rand", "peaky", "sparse", "ties"])
    u = synth.uniform01(3000 + seed, (B, 1, Hp, Wp))
    if kind == "peaky":
        u = (u ** 8).astype(np.float32)
    elif kind == "sparse":
        u = np.where(synth.uniform01(3100 + seed, (B, 1, Hp, Wp)) < np.float32(0.02), u, np.float32(0)).astype(np.float32)
    elif kind == "ties":
        lv = np.float32(int(r.choice([4, 16, 64])))
        u = (np.floor(u * lv) / lv).astype(np.float32)
    top_k = int(r.choice([0, 10, 100, 1024]))
    det_thr = float(r.choice([1.0, 1.0, 0.0, 0.6]))
    ordering = str(r.choice(["yx", "xy"]))
    sc = u.copy()
    oracle.mask_border(sc, None, pads, False, border)
    exp_nms, exp_pos, exp_idx, exp_thr, iters = oracle.detect_post(sc.copy(), top_k, radius, 0, det_thr, pads, ordering)
    cap = max(max(len(p) for p in exp_pos), 1) if (not top_k or det_thr < 1.0) else None
    budget = 8
    while True:  # radii other than 4 report too few passes instead of finishing on the device
        d = pkg.native.detect(_t(sc), top_k=top_k, radius=radius, det_thr=det_thr, pads=pads, ordering=ordering, cap=cap, nms_iters=budget)
        if not _np(d.not_converged).any():
            break
        assert radius != 4
        budget *= 4
    cnt = _np(d.counts)
    assert cnt.tolist() == [len(p) for p in exp_pos], (kind, radius, top_k, det_thr)
    for b in range(B):
        assert np.array_equal(_np(d.positions[b, :cnt[b]]), exp_pos[b])
        assert np.array_equal(_np(d.indices[b, :cnt[b]]), exp_idx[b])
    w0, w1, h0, h1 = pads
    assert np.array_equal(_np(d.nms), exp_nms[:, h0:Hp - h1, w0:Wp - w1])
