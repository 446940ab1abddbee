"""GPU tests (-m gpu), component: mnn.
SURVEY 8a rows A19-A20 (Matcher frozen branch, NearestNeighborMatcher with thresholds and its empty / zero-match quirks): mnn.hip, match_tiles.h.
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

from helpers import (la_bound, mnn_inputs, r2_mnn_inputs, sub_dict, synth)
from gpu_support import (DEV, FTOL, MNN, MNNS, TIED, _Z, _assert_feats_equal_oracle, _mnn_feats, _np, _rng, _t, pkg)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(MNN.cases))
def test_mnn_vs_golden_and_oracle(oracle, name):
    c = MNN.cases[name]
    d0, d1, k0, k1 = mnn_inputs(c)
    mm = pkg.NearestNeighborMatcher(ratio_thresh=False, distance_thresh=False, mutual_check=True)
    size = torch.tensor([260, 346])
    f0 = {"sparse_descriptors": _t(d0)[None], "sparse_positions": _t(k0)[None], "image_size": [size]}
    f1 = {"sparse_descriptors": _t(d1)[None], "sparse_positions": _t(k1)[None], "image_size": [size]}
    r = mm(f0, f1)
    exp = oracle.mnn(d0, d1)
    assert r["matches0"].dtype == torch.int64 and tuple(r["matches0"].shape) == (1, c["n"])
    for key, gk in (("matches0", "matches0"), ("matches1", "matches1"), ("matching_scores0", "mscores0"), ("matching_scores1", "mscores1")):
        assert np.array_equal(_np(r[key])[0], exp[key]), key
        assert np.array_equal(_np(r[key]), MNN[f"{name}.{gk}"]), key
    assert np.array_equal(_np(r["matched_kpts0"]), MNN[f"{name}.matched_kpts0"])
    assert np.array_equal(_np(r["matched_kpts1"]), MNN[f"{name}.matched_kpts1"])
    la = _np(r["log_assignment"])
    assert la.shape == (1, c["n"] + 1, c["m"] + 1)
    np.testing.assert_allclose(la[0], exp["log_assignment"], atol=1e-5, rtol=0)
    if f"{name}.la" in MNN:
        np.testing.assert_allclose(la, MNN[f"{name}.la"], atol=2e-5, rtol=0)


def test_mnn_empty_and_zero_match_quirks():
    mm = pkg.NearestNeighborMatcher(False, False, True)
    size = torch.tensor([260, 346])
    d = _t(synth.synth_unit_descriptors(3, 4, 256))
    k = _t(synth.uniform(4, (4, 3), 0, 100))
    empty = {"sparse_descriptors": d[:0][None], "sparse_positions": k[:0][None], "image_size": [size]}
    full = {"sparse_descriptors": d[None], "sparse_positions": k[None], "image_size": [size]}
    r = mm(empty, full)
    assert tuple(r["matches0"].shape) == (1, 0) and tuple(r["matches1"].shape) == (1, 4)
    assert tuple(r["matched_kpts0"].shape) == (0, 3) and tuple(r["log_assignment"].shape) == (1, 1, 5)


@pytest.mark.parametrize("matcher", ["MNN", "LightGlue"])
def test_ragged_keypoint_counts_in_a_batch(oracle, matcher):
    """pairs with different keypoint counts (one event sample has almost no events -> far fewer than k
    keypoints) go through the same batched launches; every pair must equal its own per-pair oracle run."""
    cfg = pkg.default_config("SP_MNN" if matcher == "MNN" else "SP_LG", event_channels=5)
    for sec in (cfg.event_extractor.vgg, cfg.image_extractor.superpointv1):
        sec.detection_top_k = 150
    model = pkg.EIM(cfg, device=DEV).eval()
    sdn = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=41)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()}, strict=False)
    for ext in (model.event_extractor.extractor, model.image_extractor.extractor):
        ext.dense_outputs = False
    B, H, W = 3, 120, 152
    ev, mask = synth.synth_events(61, B, 5, H, W)
    keep = np.zeros_like(mask[1])
    keep[:, 40:52, 60:80] = True  # sample 1: events only in a small window -> few keypoints
    mask[1] &= keep
    ev[1] *= mask[1]
    img = synth.synth_image(61, B, H, W)
    ef, imf, m = model(_t(ev), _t(img), _t(mask))
    counts = [int(p.shape[0]) for p in ef["sparse_positions"]]
    assert counts[1] < 150 and counts[0] == 150 and counts[1] > 0
    oe = oracle.extractor_forward("vgg", sub_dict(sdn, "event_extractor.extractor."), ev.copy(), mask, top_k=150)
    oi = oracle.extractor_forward("superpointv1", sub_dict(sdn, "image_extractor.extractor."), img.copy(), None, top_k=150)
    _assert_feats_equal_oracle(ef, oe)
    _assert_feats_equal_oracle(imf, oi)
    for b in range(B):
        k0, k1 = oe["sparse_positions"][b], oi["sparse_positions"][b]
        d0, d1 = oe["sparse_descriptors"][b], oi["sparse_descriptors"][b]
        if matcher == "MNN":
            r = oracle.mnn(d0, d1)
            assert np.array_equal(_np(m["matches0"][b])[0], r["matches0"])
            assert np.array_equal(_np(m["matches1"][b])[0], r["matches1"])
            np.testing.assert_allclose(_np(m["log_assignment"][b])[0], r["log_assignment"], atol=1e-5)
            cols = 3
        else:
            r = oracle.lightglue(sub_dict(sdn, "matcher.matcher."), k0, d0, k1, d1, size0=(H, W), size1=(H, W))
            assert np.array_equal(_np(m["matches0"][b])[0], r["matches0"])
            np.testing.assert_allclose(_np(m["matching_scores0"][b])[0], r["matching_scores0"], atol=FTOL)
            np.testing.assert_allclose(_np(m["log_assignment"][b])[0], r["log_assignment"], atol=la_bound("lg.d256"), rtol=0)
            cols = 2
        mk0, mk1 = oracle.matched_kpts(k0, k1, r["matches0"], cols)
        assert np.array_equal(_np(m["matched_kpts0"][b]), mk0) and np.array_equal(_np(m["matched_kpts1"][b]), mk1)
        assert tuple(m["matches0"][b].shape) == (1, len(k0)) and tuple(m["log_assignment"][b].shape) == (1, len(k0) + 1, len(k1) + 1)


@pytest.mark.parametrize("name", [n for n in MNNS if n not in TIED])
def test_mnn_thresholds_vs_reference(oracle, name):
    c = MNNS[name]
    d0, d1, k0, k1 = r2_mnn_inputs(c)
    mm = pkg.NearestNeighborMatcher(ratio_thresh=c.get("ratio") or False, distance_thresh=c.get("dist") or False, mutual_check=True)
    r = mm(*_mnn_feats(d0, d1, k0, k1))
    assert np.array_equal(_np(r["matches0"]), _Z[f"{name}.matches0"])
    assert np.array_equal(_np(r["matches1"]), _Z[f"{name}.matches1"])
    assert np.array_equal(_np(r["matched_kpts0"]), _Z[f"{name}.matched_kpts0"])
    assert np.array_equal(_np(r["matched_kpts1"]), _Z[f"{name}.matched_kpts1"])
    exp = oracle.mnn_thresh(d0, d1, c.get("ratio"), c.get("dist"))
    assert np.array_equal(_np(r["matching_scores0"])[0], exp["matching_scores0"])


@pytest.mark.parametrize("name", TIED)
def test_mnn_exact_ties_first_index_rule(oracle, name):
    """exact ties in sim: the kernels implement the build's first-index rule bit for bit like the oracle (the
    reference's own pick among equals is an artefact of torch's partial sort, see test_r2_golden_cpu.py)."""
    c = MNNS[name]
    d0, d1, k0, k1 = r2_mnn_inputs(c)
    mm = pkg.NearestNeighborMatcher(ratio_thresh=c.get("ratio") or False, distance_thresh=False, mutual_check=True)
    r = mm(*_mnn_feats(d0, d1, k0, k1))
    exp = oracle.mnn_thresh(d0, d1, c.get("ratio"), None) if c.get("ratio") else oracle.mnn(d0, d1, want_la=False)
    assert np.array_equal(_np(r["matches0"])[0], exp["matches0"])
    assert np.array_equal(_np(r["matches1"])[0], exp["matches1"])
    assert int((exp["matches0"] > -1).sum()) >= 1


def test_mnn_thresholds_in_a_ragged_batch(oracle):
    """the thresholded matcher inside a batch with different counts per pair (device-side counts)"""
    nat = pkg.native
    cases = [MNNS["ratio"], MNNS["both"], MNNS["dist"]]
    ins = [r2_mnn_inputs(c) for c in cases]
    cap0, cap1, D = 300, 300, 64
    B = len(ins)
    d0 = np.zeros((B, cap0, 256), np.float32)
    d1 = np.zeros((B, cap1, 256), np.float32)
    n, m = [], []
    for b, (a0, a1, _, _) in enumerate(ins):
        d0[b, :a0.shape[0], :a0.shape[1]] = a0
        d1[b, :a1.shape[0], :a1.shape[1]] = a1
        n.append(a0.shape[0])
        m.append(a1.shape[0])
    r = nat.mnn(_t(d0), torch.tensor(n, dtype=torch.int32, device=DEV), _t(d1), torch.tensor(m, dtype=torch.int32, device=DEV),
                want_la=False, ratio_thresh=0.9, distance_thresh=0.8)
    for b in range(B):
        exp = oracle.mnn_thresh(d0[b, :n[b]], d1[b, :m[b]], 0.9, 0.8)
        assert np.array_equal(_np(r.matches0)[b, :n[b]], exp["matches0"])
        assert np.array_equal(_np(r.matches1)[b, :m[b]], exp["matches1"])
        assert (_np(r.matches0)[b, n[b]:] == -1).all()


def test_ratio_threshold_single_candidate_raises():
    d0, d1, k0, k1 = r2_mnn_inputs(MNNS["dup"])
    mm = pkg.NearestNeighborMatcher(ratio_thresh=0.8, distance_thresh=False, mutual_check=True)
    with pytest.raises(RuntimeError, match="out of range"):
        mm(*_mnn_feats(d0[:5], d1[:1], k0[:5], k1[:1]))


def test_matcher_uses_edited_feature_lists(oracle):
    """a caller may filter feats['sparse_positions'] / ['sparse_descriptors'] between the extractor and
    Matcher(feats0, feats1) (legal with the reference, which reads the lists); the hidden device batch must not be
    used then."""
    cfg = pkg.default_config("SP_MNN", event_channels=5)
    for sec in (cfg.event_extractor.vgg, cfg.image_extractor.superpointv1):
        sec.detection_top_k = 64
    model = pkg.EIM(cfg, device=DEV).eval()
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=5)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    ev, mask = synth.synth_events(9, 2, 5, 90, 122)
    img = synth.synth_image(9, 2, 90, 122)
    ef = model.event_extractor(_t(ev), _t(mask))
    imf = model.image_extractor(_t(img))
    full = model.matcher(ef, imf)
    for b in range(2):
        exp = oracle.mnn(_np(ef["sparse_descriptors"][b]), _np(imf["sparse_descriptors"][b]), want_la=False)
        assert np.array_equal(_np(full["matches0"][b])[0], exp["matches0"])
    # keep every second keypoint of image 0 on the event side (new tensors, new lengths)
    keep = [ef["sparse_positions"][b].shape[0] for b in range(2)]
    ef["sparse_positions"] = [p[::2].contiguous() for p in ef["sparse_positions"]]
    ef["sparse_descriptors"] = [d[::2].contiguous() for d in ef["sparse_descriptors"]]
    cut = model.matcher(ef, imf)
    for b in range(2):
        n = (keep[b] + 1) // 2
        assert tuple(cut["matches0"][b].shape) == (1, n)
        exp = oracle.mnn(_np(ef["sparse_descriptors"][b]), _np(imf["sparse_descriptors"][b]), want_la=False)
        assert np.array_equal(_np(cut["matches0"][b])[0], exp["matches0"])
        assert np.array_equal(_np(cut["matches1"][b])[0], exp["matches1"])
    # a shortened VIEW of the original rows (same storage, fewer rows) must be honoured too
    imf["sparse_positions"] = [p[:10] for p in imf["sparse_positions"]]
    imf["sparse_descriptors"] = [d[:10] for d in imf["sparse_descriptors"]]
    cut2 = model.matcher(ef, imf)
    for b in range(2):
        exp = oracle.mnn(_np(ef["sparse_descriptors"][b]), _np(imf["sparse_descriptors"][b]), want_la=False)
        assert tuple(cut2["matches1"][b].shape) == (1, 10)
        assert np.array_equal(_np(cut2["matches0"][b])[0], exp["matches0"])


@pytest.mark.parametrize("seed", list(range(10)))
def test_random_mnn_batches(oracle, seed):
    r = _rng(4000 + seed)
    B = int(r.integers(1, 6))
    D = int(r.choice([64, 128, 256]))
    cap0, cap1 = int(r.integers(1, 400)), int(r.integers(1, 400))
    n = [int(r.integers(0, cap0 + 1)) for _ in range(B)]
    m = [int(r.integers(0, cap1 + 1)) for _ in range(B)]
    n[0], m[0] = cap0, cap1
    d0 = np.stack([synth.synth_unit_descriptors(5000 + seed * 10 + b, cap0, D) for b in range(B)])
    d1 = np.stack([synth.synth_unit_descriptors(6000 + seed * 10 + b, cap1, D) for b in range(B)])
    for b in range(B):  # plant real matches
        s = min(n[b], m[b]) // 2
        d1[b, :s] = d0[b, :s]
    use_thr = seed % 3 == 0
    rt, dt = (0.9, 0.8) if use_thr else (None, None)
    res = pkg.native.mnn(_t(d0), torch.tensor(n, dtype=torch.int32, device=DEV), _t(d1), torch.tensor(m, dtype=torch.int32, device=DEV),
                         want_la=not use_thr, ratio_thresh=rt, distance_thresh=dt)
    for b in range(B):
        if n[b] == 0 or m[b] == 0:
            assert (_np(res.matches0)[b] == -1).all() and (_np(res.matches1)[b] == -1).all()
            continue
        if use_thr and (n[b] < 2 or m[b] < 2):
            continue  # the reference's topk(2) raises there; the batched kernel lets the ratio test pass
        exp = oracle.mnn_thresh(d0[b, :n[b]], d1[b, :m[b]], rt, dt) if use_thr else oracle.mnn(d0[b, :n[b]], d1[b, :m[b]])
        assert np.array_equal(_np(res.matches0)[b, :n[b]], exp["matches0"]), (b, n[b], m[b])
        assert np.array_equal(_np(res.matches1)[b, :m[b]], exp["matches1"])
        assert (_np(res.matches0)[b, n[b]:] == -1).all()
        if not use_thr:
            np.testing.assert_allclose(_np(res.la)[b, :n[b] + 1, :m[b] + 1], exp["log_assignment"], atol=1e-5, rtol=0)
