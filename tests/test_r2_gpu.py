"""Round-2 GPU parity tests (-m gpu): the gaps VERDICT r1 listed.

* BASELINE.json's own batch sizes: sampled pairs (first / middle / last) of B=32 SP+MNN, B=32 SiLK+MNN and
  B=64 SP+LightGlue against per-pair oracle runs (persistent XCD-aware workgroups, blockIdx.z = pair and the
  3-workgroups-per-CU attention only show their indexing at these sizes);
* round-2 fixtures generated from the reference (tests/golden/r2.npz): tie maps with survivors, find_nn's
  ratio / distance thresholds, Repeatability; exact-tie descriptor inputs against the oracle's first-index rule;
* the documented drop-in (`install_as_core()`) driven through the reference scripts' own import lines;
* a parent-level load_state_dict after a forward really changes the weights the kernels use.
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

from helpers import GOLDEN, close_and_record, la_bound, load_pkg, r2_mnn_inputs, record_flips, rep_inputs, sub_dict, synth, tie_map

pytestmark = pytest.mark.gpu
pkg = load_pkg()
DEV = "cuda:0"
FTOL = 1e-4
# log_assignment: bounded by a multiple of the reference's OWN summation-order / rounding noise on the matching fixture
# (helpers.la_bound, tests/golden/lgcal.npz), not by a hand-picked number

_Z = np.load(os.path.join(GOLDEN, "r2.npz"))
_META = json.loads(bytes(_Z["meta"]).decode())
TIES = {c["name"]: c for c in _META["tie_cases"]}
MNNS = {c["name"]: c for c in _META["mnn_cases"]}
REPS = {c["name"]: c for c in _META["rep_cases"]}
TIED = ("alleq", "dup", "zero", "ratio_dup")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _np(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module", autouse=True)
def _require_gpu():
    assert torch.cuda.is_available(), "these tests need a HIP device"
    yield
    torch.cuda.synchronize()


# ------------------------------------------------------------------ BASELINE batch sizes vs per-pair oracle
def _bench_like_model(cfg_name, seed=11):
    cfg = pkg.default_config(cfg_name, event_channels=5)
    model = pkg.EIM(cfg, device=DEV).eval()
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    for ext in (model.event_extractor.extractor, model.image_extractor.extractor):
        ext.dense_outputs = False
    return cfg, model, sd


def _calibrate(model, sd, ev, img, mask):
    """bench.py's descriptor-bias calibration (distinct descriptors -> tens of matches per pair)"""
    ef, imf, _ = model(ev, img.clone(), mask)
    msd = model.state_dict()
    over = {}
    for prefix, feats in (("event_extractor.extractor.", ef), ("image_extractor.extractor.", imf)):
        mean = feats["raw_descriptors"].mean(dim=(0, 2, 3))
        key = [k for k in msd if k.startswith(prefix) and (k.endswith("convDb.bias") or k.endswith("_desH2.1.bias"))]
        assert len(key) == 1
        over[key[0]] = (msd[key[0]] - mean).detach().cpu()
    model.load_state_dict(over, strict=False)
    for k, v in over.items():
        sd[k] = v.numpy()


def _oracle_pair(oracle, cfg, sd, ev, mask, img, b):
    et, it = cfg.event_extractor.type, cfg.image_extractor.type
    oe = oracle.extractor_forward(et, sub_dict(sd, "event_extractor.extractor."), ev[b:b + 1].copy(), mask[b:b + 1], top_k=1024,
                                  scale=cfg.event_extractor[et].descriptor_scale_factor)
    oi = oracle.extractor_forward(it, sub_dict(sd, "image_extractor.extractor."), img[b:b + 1].copy(), None, top_k=1024,
                                  scale=cfg.image_extractor[it].descriptor_scale_factor)
    return oe, oi


@pytest.mark.parametrize("cfg_name,B", [("SP_MNN", 32), ("SiLK_MNN", 32)])
def test_baseline_batch_sampled_pairs_vs_oracle_mnn(oracle, cfg_name, B):
    """configs[1] / configs[2] at their real batch size: pairs 0, B/2, B-1 bit-equal to per-pair oracle runs
    (keypoints + scores, descriptors, match indices, matched keypoints)."""
    cfg, model, sd = _bench_like_model(cfg_name)
    ev, mask = synth.synth_events(10_000, B, 5)
    img = synth.synth_image(10_000, B)
    ev_t, mask_t, img_t = _t(ev), _t(mask), _t(img)
    _calibrate(model, sd, ev_t, img_t, mask_t)
    ef, imf, m = model(ev_t, img_t.clone(), mask_t)
    nmatch = []
    for b in (0, B // 2, B - 1):
        oe, oi = _oracle_pair(oracle, cfg, sd, ev, mask, img, b)
        for got, exp in ((ef, oe), (imf, oi)):
            assert np.array_equal(_np(got["sparse_positions"][b]), exp["sparse_positions"][0]), f"pair {b}: keypoints"
            assert np.array_equal(_np(got["sparse_descriptors"][b]), exp["sparse_descriptors"][0]), f"pair {b}: descriptors"
        r = oracle.mnn(oe["sparse_descriptors"][0], oi["sparse_descriptors"][0], want_la=False)
        assert np.array_equal(_np(m["matches0"][b])[0], r["matches0"]), f"pair {b}: matches0"
        assert np.array_equal(_np(m["matches1"][b])[0], r["matches1"]), f"pair {b}: matches1"
        mk0, mk1 = oracle.matched_kpts(oe["sparse_positions"][0], oi["sparse_positions"][0], r["matches0"], 3)
        assert np.array_equal(_np(m["matched_kpts0"][b]), mk0) and np.array_equal(_np(m["matched_kpts1"][b]), mk1)
        nmatch.append(int((r["matches0"] > -1).sum()))
    assert min(nmatch) >= 5, f"calibrated descriptors should give real matches, got {nmatch}"


@pytest.mark.parametrize("cfg_name", ["SP_MNN", "SP_LG"])
def test_pair_without_events_gives_the_reference_empty_dict_and_leaves_the_batch_alone(cfg_name):
    """A pair whose events mask is empty has no event keypoints (score[~mask] = 0, EventExtractors.py:561-562): the matcher
    returns the reference's empty-input dict for THAT pair (MNN.py:63-86 / lightglue.py:572-591: matches of length 0 / m, no
    matched keypoints, zero log_assignment [1,1,m+1]) and the other pairs of the batch are exactly what they are alone."""
    cfg, model, sd = _bench_like_model(cfg_name)
    B = 3
    ev, mask = synth.synth_events(777, B, 5)
    img = synth.synth_image(777, B)
    ev[1] = 0.0
    mask[1] = False
    ef, imf, m = model(_t(ev), _t(img), _t(mask))
    assert ef["sparse_positions"][1].shape == (0, 3) and ef["sparse_descriptors"][1].shape == (0, 256)
    mi = imf["sparse_positions"][1].shape[0]
    assert mi > 0
    assert tuple(m["matches0"][1].shape) == (1, 0) and tuple(m["matches1"][1].shape) == (1, mi)
    assert bool((m["matches1"][1] == -1).all()) and float(m["matching_scores1"][1].abs().max()) == 0.0
    assert tuple(m["matched_kpts0"][1].shape) == (0, 3) and tuple(m["matched_kpts1"][1].shape) == (0, 3)
    la = m["log_assignment"][1]
    assert tuple(la.shape) == (1, 1, mi + 1) and float(la.abs().max()) == 0.0
    for b in (0, 2):
        ef1, imf1, m1 = model(_t(ev[b:b + 1]), _t(img[b:b + 1]), _t(mask[b:b + 1]))
        assert torch.equal(ef1["sparse_positions"][0], ef["sparse_positions"][b])
        assert torch.equal(imf1["sparse_descriptors"][0], imf["sparse_descriptors"][b])
        assert torch.equal(m1["matches0"][0], m["matches0"][b])
        assert torch.equal(m1["matched_kpts0"][0], m["matched_kpts0"][b])


def test_other_geometry_vga_16_bins_vs_oracle(oracle):
    """Nothing is specialised to 346x260 / 5 bins: one 640x480 pair with the reference's shipped 16 event bins (N = 307,200 score
    pixels: the generic selection path, 60x80 heads, other conv tile choices) bit-equal to the oracle end to end."""
    cfg = pkg.default_config("SP_MNN", event_channels=16)
    model = pkg.EIM(cfg, device=DEV).eval()
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=17)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    for ext in (model.event_extractor.extractor, model.image_extractor.extractor):
        ext.dense_outputs = False
    H, W = 480, 640
    ev, mask = synth.synth_events(4321, 1, 16, H, W)
    img = synth.synth_image(4321, 1, H, W)
    ef, imf, m = model(_t(ev), _t(img), _t(mask))
    oe = oracle.extractor_forward("vgg", sub_dict(sd, "event_extractor.extractor."), ev.copy(), mask, top_k=1024)
    oi = oracle.extractor_forward("superpointv1", sub_dict(sd, "image_extractor.extractor."), img.copy(), None, top_k=1024)
    for got, exp in ((ef, oe), (imf, oi)):
        assert np.array_equal(_np(got["sparse_positions"][0]), exp["sparse_positions"][0])
        assert np.array_equal(_np(got["sparse_descriptors"][0]), exp["sparse_descriptors"][0])
        assert np.array_equal(_np(got["score"]), exp["score"])
    r = oracle.mnn(oe["sparse_descriptors"][0], oi["sparse_descriptors"][0], want_la=False)
    assert np.array_equal(_np(m["matches0"][0])[0], r["matches0"])
    assert 0 < ef["sparse_positions"][0].shape[0] <= 1024 and tuple(ef["score"].shape) == (1, 1, H, W)


def _calibrate_lightglue(model, sd, ef, imf):
    """synth.lightglue_calibration from the final descriptors of pair 0 (the same rule as tests/golden/lgcal.npz and bench.py)"""
    one = lambda f: {"sparse_positions": f["sparse_positions"][0][None], "sparse_descriptors": f["sparse_descriptors"][0][None],  # noqa: E731
                     "image_size": [f["image_size"][0]]}
    r = model.matcher.matcher(one(ef), one(imf))
    x = np.concatenate([_np(r["ref_descriptors0"])[0, 0], _np(r["ref_descriptors1"])[0, 0]], 0)
    over, _ = synth.lightglue_calibration(sd, x, prefix="matcher.matcher.")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in over.items()}, strict=False)
    sd.update(over)


def test_baseline_batch_sampled_pairs_vs_oracle_lightglue(oracle):
    """configs[3]: B=64 SP+LightGlue on "same scene" pairs with a calibrated assignment head (hundreds of confident matches per
    pair, synth.twin_overrides / lightglue_calibration): extractor outputs bit-equal, match assignments equal (flips are counted
    and reported; target 0), matching scores to 1e-4, log_assignment to a multiple of the reference's own noise floor."""
    B = 64
    cfg, model, sd = _bench_like_model("SP_LG")
    tw = synth.twin_overrides(sd)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in tw.items()}, strict=False)
    sd.update(tw)
    ev, mask = synth.synth_events(10_000, B, 5)
    img = synth.synth_image(10_000, B)
    ev = synth.twin_events(ev, img)
    ev_t, mask_t, img_t = _t(ev), _t(mask), _t(img)
    _calibrate(model, sd, ev_t, img_t, mask_t)
    ef, imf, _ = model(ev_t, img_t.clone(), mask_t)
    _calibrate_lightglue(model, sd, ef, imf)
    ef, imf, m = model(ev_t, img_t.clone(), mask_t)
    bound = la_bound("sp_lg_twin.0")
    for b in (0, B // 2, B - 1):
        oe, oi = _oracle_pair(oracle, cfg, sd, ev, mask, img, b)
        for got, exp in ((ef, oe), (imf, oi)):
            assert np.array_equal(_np(got["sparse_positions"][b]), exp["sparse_positions"][0]), f"pair {b}: keypoints"
            assert np.array_equal(_np(got["sparse_descriptors"][b]), exp["sparse_descriptors"][0]), f"pair {b}: descriptors"
        r = oracle.lightglue(sub_dict(sd, "matcher.matcher."), oe["sparse_positions"][0], oe["sparse_descriptors"][0],
                             oi["sparse_positions"][0], oi["sparse_descriptors"][0])
        g0, e0 = _np(m["matches0"][b])[0], np.asarray(r["matches0"]).reshape(-1)
        assert int((e0 > -1).sum()) >= 100, f"pair {b}: the calibrated workload should match hundreds of keypoints, got {int((e0 > -1).sum())}"
        record_flips("B64 sp_lg (same scene) matches0 vs oracle", g0, e0, r["log_assignment"])
        # flash-style attention sums in a different order than the oracle: an assignment may differ ONLY at rows of the
        # allow-list built from the oracle's own decision margins (best minus second best of the row, or of the chosen column,
        # inside the same-input log_assignment bound); no count budget
        la = np.asarray(r["log_assignment"])[:-1, :-1]
        top2r = np.sort(la, axis=1)[:, -2:]
        top2c = np.sort(la, axis=0)[-2:, :]
        row_gap = top2r[:, 1] - top2r[:, 0]
        col_gap = top2c[1] - top2c[0]
        allow = {int(i) for i in np.nonzero((row_gap < bound) | (col_gap[np.argmax(la, axis=1)] < bound))[0]}
        bad = np.nonzero(g0 != e0)[0]
        assert {int(i) for i in bad} <= allow, f"pair {b}: rows {sorted({int(i) for i in bad} - allow)} differ outside the oracle's near-tie rows"
        gla = _np(m["log_assignment"][b])[0] if m["log_assignment"][b] is not None else None
        if gla is not None:
            close_and_record("B64 sp_lg (same scene) log_assignment vs oracle", gla[::53, ::47], r["log_assignment"][::53, ::47], atol=bound)
        es = np.asarray(r["matching_scores0"]).reshape(-1)
        assert es.max() > 0.9 and ((es > 0.1) & (es < 0.9)).sum() >= 100
        close_and_record("B64 sp_lg (same scene) matching_scores0 vs oracle", _np(m["matching_scores0"][b])[0], es, atol=1e-4)


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


# ------------------------------------------------------------------ r2 fixtures: MNN thresholds / exact ties
def _mnn_feats(d0, d1, k0, k1):
    size = torch.tensor([260, 346])
    f0 = {"sparse_descriptors": _t(d0)[None], "sparse_positions": _t(k0)[None], "image_size": [size]}
    f1 = {"sparse_descriptors": _t(d1)[None], "sparse_positions": _t(k1)[None], "image_size": [size]}
    return f0, f1


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


# ------------------------------------------------------------------ r2 fixtures: Repeatability
@pytest.mark.parametrize("name", list(REPS))
def test_repeatability_class_vs_reference(name):
    from importlib import import_module
    km = import_module(pkg.__name__ + ".core.metrics.keypoints_metrics")
    c = REPS[name]
    p0, p1 = rep_inputs(c)
    Hm = torch.eye(3) if c["hom"] is None else torch.tensor(c["hom"], dtype=torch.float32).reshape(3, 3)
    got = []
    for t in (1, 3):
        d = km.Repeatability(f"repeatability@{t}", distance_thresh=t, ordering=c["ordering"]).update_one(
            _t(p0), _t(p1), (260, 346), (260, 346), Hm.to(DEV))
        got.append(d.get(f"repeatability@{t}", float("nan")))
    np.testing.assert_allclose(np.array(got), _Z[f"{name}.values"], atol=1e-7, rtol=1e-6)
    # update_batch = mean over the pairs that produced a value (keypoints_metrics.py:134-157)
    R = km.Repeatability("r", distance_thresh=3, ordering=c["ordering"])
    both = R.update_batch([_t(p0), _t(p0)], [_t(p1), _t(p1)], (260, 346), (260, 346), torch.stack([Hm, Hm]).to(DEV))
    assert abs(both["r"] - float(_Z[f"{name}.values"][1])) < 1e-6
    assert km.Repeatability("r", 3).update_one(_t(p0[:0]), _t(p1[:0]), (260, 346), (260, 346), Hm.to(DEV)) == {}
    # the reference omits the entry when no keypoint of EITHER image survives keep_true_points
    # (original_num + warped_num == 0, keypoints_metrics.py:126-128), not only when the inputs are empty: a translation by
    # 1000 px moves every point out of both frames; update_batch then averages the other pairs only
    far = torch.tensor([[1.0, 0.0, 1000.0], [0.0, 1.0, 1000.0], [0.0, 0.0, 1.0]])
    assert km.Repeatability("r", 3, ordering=c["ordering"]).update_one(_t(p0), _t(p1), (260, 346), (260, 346), far.to(DEV)) == {}
    mixed = R.update_batch([_t(p0), _t(p0)], [_t(p1), _t(p1)], (260, 346), (260, 346), torch.stack([Hm, far]).to(DEV))
    assert abs(mixed["r"] - float(_Z[f"{name}.values"][1])) < 1e-6


# ------------------------------------------------------------------ the documented drop-in: install_as_core()
def test_install_as_core_runs_reference_style_imports_and_a_forward(oracle):
    """INTEGRATION.md's drop-in: after install_as_core() the import lines of the reference's evaluation script
    (test_events-image_same-time.py:13,19-25,30-44) resolve to the native build; one forward through them."""
    saved = {k: v for k, v in sys.modules.items() if k == "core" or k.startswith("core.")}
    try:
        pkg.install_as_core()
        ns = {}
        exec("from core.modules import build_model\n"
             "from core.modules.EIM import EIM\n"
             "from core.metrics.keypoints_metrics import Repeatability, ValidDescriptorsDistance\n"
             "from core.metrics.matching_metrics import (MeanMatchingAccuracy, MatchingRatio, HomographyEstimation,\n"
             "                                           RelativePoseEstimation, compute_auc)\n"
             "from core.modules.utils.detector_util import (logits_to_prob, depth_to_space, prob_map_to_points_map,\n"
             "                                              prob_map_to_positions_with_prob, get_dense_positions)\n"
             "from core.modules.utils.descriptor_util import (normalize_descriptors, get_dense_descriptors,\n"
             "                                                sparsify_full_resolution_descriptors,\n"
             "                                                sparsify_low_resolution_descriptors, upsample_descriptors)\n", ns)
        assert ns["EIM"] is pkg.EIM
        with pytest.raises(NotImplementedError, match="OpenCV"):
            ns["HomographyEstimation"]("HE")
        assert abs(ns["compute_auc"]([0.5, 1.2, 3.0, 7.0, float("inf"), 2.2], [5])["5"] - 0.5839999961853027) < 1e-12
        cfg = pkg.default_config("SP_MNN", event_channels=5)
        for sec in (cfg.event_extractor.vgg, cfg.image_extractor.superpointv1):
            sec.detection_top_k = 64
        model = ns["build_model"](cfg, DEV, None).eval()
        sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=5)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        ev, mask = synth.synth_events(9, 1, 5, 90, 122)
        img = synth.synth_image(9, 1, 90, 122)
        ef, imf, m = model(_t(ev), _t(img), _t(mask))
        oe = oracle.extractor_forward("vgg", sub_dict(sd, "event_extractor.extractor."), ev.copy(), mask, top_k=64)
        assert np.array_equal(_np(ef["sparse_positions"][0]), oe["sparse_positions"][0])
        # the harness-style metric calls of the script (:236-262) on this forward
        vdd = ns["ValidDescriptorsDistance"]("VDD", [1, 3]).update_one(ef["sparse_positions"][0], imf["sparse_positions"][0],
                                                                      ef["sparse_descriptors"][0], imf["sparse_descriptors"][0],
                                                                      (90, 122), (90, 122), torch.eye(3, device=DEV))
        rep = ns["Repeatability"]("rep@3", distance_thresh=3, ordering="yx").update_one(
            ef["sparse_positions"][0][:, :2].contiguous(), imf["sparse_positions"][0][:, :2].contiguous(), (90, 122), (90, 122),
            torch.eye(3, device=DEV))
        assert abs(rep["rep@3"] - vdd["VDD_Repeatability@3"]) < 1e-9
    finally:
        for k in [k for k in sys.modules if k == "core" or k.startswith("core.")]:
            del sys.modules[k]
        sys.modules.update(saved)


# ------------------------------------------------------------------ parent load_state_dict after a forward
def test_parent_load_state_dict_after_forward_uses_the_new_weights(oracle):
    cfg = pkg.default_config("SP_MNN", event_channels=5)
    for sec in (cfg.event_extractor.vgg, cfg.image_extractor.superpointv1):
        sec.detection_top_k = 64
    model = pkg.EIM(cfg, device=DEV).eval()
    keys = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    ev, mask = synth.synth_events(9, 1, 5, 90, 122)
    img = synth.synth_image(9, 1, 90, 122)
    outs = []
    for seed in (5, 6):  # second load goes through EIM.load_state_dict AFTER a forward built the native images
        sd = synth.synth_state_dict(keys, seed=seed)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        ef, imf, m = model(_t(ev), _t(img), _t(mask))
        oe = oracle.extractor_forward("vgg", sub_dict(sd, "event_extractor.extractor."), ev.copy(), mask, top_k=64)
        oi = oracle.extractor_forward("superpointv1", sub_dict(sd, "image_extractor.extractor."), img.copy(), None, top_k=64)
        assert np.array_equal(_np(ef["sparse_descriptors"][0]), oe["sparse_descriptors"][0]), f"seed {seed}: stale event weights"
        assert np.array_equal(_np(imf["sparse_descriptors"][0]), oi["sparse_descriptors"][0]), f"seed {seed}: stale image weights"
        outs.append(_np(ef["logits"]).copy())
    assert not np.array_equal(outs[0], outs[1])
    # in-place edit of one parameter (no load_state_dict at all) is picked up too
    with torch.no_grad():
        model.image_extractor.extractor.convPb.bias.add_(0.25)
    sd["image_extractor.extractor.convPb.bias"] = sd["image_extractor.extractor.convPb.bias"] + np.float32(0.25)
    ef, imf, m = model(_t(ev), _t(img), _t(mask))
    oi = oracle.extractor_forward("superpointv1", sub_dict(sd, "image_extractor.extractor."), img.copy(), None, top_k=64)
    assert np.array_equal(_np(imf["logits"]), oi["logits"])


def test_wrong_dtype_inputs_raise_or_are_cast():
    from importlib import import_module
    du = import_module(pkg.__name__ + ".core.modules.utils.detector_util")
    s32 = _t(synth.uniform01(3, (1, 1, 40, 48)))
    ref = du.fast_nms(s32.clone(), 4)
    for dt in (torch.float64, torch.float16):
        got = du.fast_nms(s32.to(dt), 4)  # the reference helpers accept any float dtype: cast, never misread
        if dt == torch.float64:
            assert torch.equal(got, ref)
        else:
            assert got.dtype == torch.float32 and got.shape == ref.shape
    with pytest.raises(TypeError, match="float32"):
        pkg.native.detect(s32.double(), top_k=10, radius=4, det_thr=1.0)
    with pytest.raises(TypeError, match="int32"):
        pkg.native.mnn(torch.zeros(1, 8, 64, device=DEV), torch.tensor([8], device=DEV), torch.zeros(1, 8, 64, device=DEV),
                       torch.tensor([8], device=DEV))


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


def test_handle_level_extract_equals_op_level_calls():
    """einx_extract (one call per network) against the same network run layer by layer through the op-level ABI."""
    nat = pkg.native
    cfg = pkg.default_config("SP_MNN", event_channels=5)
    model = pkg.EIM(cfg, device=DEV).eval()
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=3)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    ev, mask = synth.synth_events(21, 3, 5, 100, 130)
    ext = model.event_extractor.extractor
    ext.dense_outputs = False
    bf = ext.extract_batched(_t(ev), _t(mask))
    eng = ext.engine()
    pads = nat.padder_pads(100, 130, 8)
    Hp, Wp = 100 + pads[2] + pads[3], 130 + pads[0] + pads[1]
    t = _t(ev)
    for i, layer in enumerate(eng.backbone):
        t = layer(t, fold=(pads[2], pads[0], Hp, Wp) if i == 0 else None)
    assert torch.equal(t, bf.feats)
    d = t
    for layer in eng.det_head:
        d = layer(d)
    assert torch.equal(d, bf.logits)
    r = t
    for layer in eng.desc_head:
        r = layer(r)
    assert torch.equal(r, bf.raw)
    coarse, raw_cl = nat.normalize_map(r, 1.0, want_cl=True)
    prob, score = nat.score_map(d, _t(mask), pads, dilate=True, border=4)
    assert torch.equal(prob, bf.prob) and torch.equal(score, bf.score) and torch.equal(coarse, bf.coarse)
    det = nat.detect(score, top_k=1024, radius=4, det_thr=1.0, pads=pads)
    assert det.cap == bf.det.cap
    assert torch.equal(det.counts, bf.det.counts) and torch.equal(det.nms, bf.det.nms)
    n = int(det.counts.min())
    assert torch.equal(det.positions[:, :n], bf.det.positions[:, :n])
    sp = nat.desc_sample(r, det.indices, det.counts, (Hp, Wp), bilinear=True, scale=1.0, raw_cl=raw_cl)
    assert torch.equal(sp[:, :n], bf.sparse_desc[:, :n])


def test_forward_stream_equals_forward():
    """EIM.forward_stream (batches in flight) returns, in order, exactly what EIM.forward returns batch by batch."""
    cfg = pkg.default_config("SP_MNN", event_channels=5)
    for sec in (cfg.event_extractor.vgg, cfg.image_extractor.superpointv1):
        sec.detection_top_k = 128
    model = pkg.EIM(cfg, device=DEV).eval()
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=5)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    batches = []
    for i in range(4):
        ev, mask = synth.synth_events(40 + i, 3, 5, 120, 160)
        batches.append((_t(ev), synth.synth_image(40 + i, 3, 120, 160), _t(mask)))
    ref = [model(ev, _t(img), mask) for ev, img, mask in batches]
    got = list(model.forward_stream(((ev, _t(img), mask) for ev, img, mask in batches), depth=2))
    assert len(got) == len(ref)
    for (e0, i0, m0), (e1, i1, m1) in zip(ref, got):
        for a, b in ((e0, e1), (i0, i1)):
            for key in ("score", "nms", "logits", "raw_descriptors"):
                assert torch.equal(a[key], b[key]), key
            for x, y in zip(a["sparse_positions"], b["sparse_positions"]):
                assert torch.equal(x, y)
            for x, y in zip(a["sparse_descriptors"], b["sparse_descriptors"]):
                assert torch.equal(x, y)
        for key in ("matches0", "matches1", "matched_kpts0", "matched_kpts1", "log_assignment"):
            for x, y in zip(m0[key], m1[key]):
                assert torch.equal(x, y), key
    ev, img, mask = batches[0]
    assert [len(list(model.forward_stream(iter([(ev, _t(img), mask)]), depth=d))) for d in (1, 3)] == [1, 1]


def test_bench_one_rank_through_the_launcher_uses_rccl():
    """`python bench.py --gpus 1 --spawn`: the launcher path of `--gpus N` with one rank on this box's one GPU --
    a fresh rank process, init_process_group("nccl") = RCCL, the metric all-reduce, one JSON line from rank 0."""
    import subprocess
    from helpers import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--steps", "2", "--warmup", "0",
                        "--no-cpu-baseline", "--no-extras", "--batch", "4"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["rccl"]["world"] == 1 and d["rccl"]["backend"] == "nccl"
    assert d["value"] > 0 and d["config"]["pairs_per_gpu_per_step"] == 4 and d["roofline"]["frac"] > 0


# ------------------------------------------------------------------ padding=0 networks (cell 1)
PAD0 = {c["name"]: c for c in _META["pad0_cases"]}


@pytest.mark.parametrize("name", list(PAD0))
def test_padding0_networks_vs_oracle_and_reference(oracle, name):
    """SiLKModel(padding=0) / VGGExtractorNP(padding=0): un-padded convolutions + `mapping_positions` (+9)
    (silk_extractor.py:142-152, EventExtractors.py:319-329): bit-equal to the oracle, 1e-4 to the reference's arithmetic."""
    from helpers import state_dict_for
    from test_oracle_golden import _check_feats
    from test_r2_golden_cpu import _G
    c = PAD0[name]
    cfg = pkg.configs.to_attr(c["cfg"])
    model = pkg.EIM(cfg, device=DEV).eval()
    sd = state_dict_for(c)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    ev, mask = synth.synth_events(c["iseed"], c["B"], c["ce"], c["H"], c["W"])
    img = synth.synth_image(c["iseed"], c["B"], c["H"], c["W"])
    ef = model.event_extractor(_t(ev), None)
    img_t = _t(img)
    imf = model.image_extractor(img_t)
    assert np.array_equal(_np(img_t), img)  # SiLK leaves the caller's image untouched
    oe = oracle.extractor_forward("vgg_np", sub_dict(sd, "event_extractor.extractor."), ev.copy(), None, top_k=c["k"], scale=1.41,
                                  padding=0, dense=True)
    oi = oracle.extractor_forward("silk", sub_dict(sd, "image_extractor.extractor."), img.copy(), None, top_k=c["k"], scale=1.41,
                                  padding=0, dense=True)
    for got, exp in ((ef, oe), (imf, oi)):
        for k in ("backbone_feats", "logits", "raw_descriptors", "probability", "score", "nms", "normalized_descriptors"):
            assert np.array_equal(_np(got[k]), exp[k]), k
        for b in range(c["B"]):
            assert np.array_equal(_np(got["sparse_positions"][b]), exp["sparse_positions"][b])
            assert np.array_equal(_np(got["sparse_descriptors"][b]), exp["sparse_descriptors"][b])
        assert tuple(got["score"].shape) == (c["B"], 1, c["H"] - 18, c["W"] - 18)
        assert [tuple(_np(s)) for s in got["image_size"]] == [(c["H"], c["W"])] * c["B"]
    as_np = lambda f: {k: (_np(v) if torch.is_tensor(v) else [_np(t) for t in v]) for k, v in f.items()}  # noqa: E731
    _check_feats(f"{name}.ev", as_np(ef), _G())
    _check_feats(f"{name}.im", as_np(imf), _G())
    np.testing.assert_allclose(_np(ef["dense_positions"][0])[::97], _Z[f"{name}.ev.dense_positions_probe"], atol=1e-6)
    np.testing.assert_allclose(_np(imf["dense_positions"][0])[::97], _Z[f"{name}.im.dense_positions_probe"], atol=1e-6)
    assert float(ef["sparse_positions"][0][:, :2].min()) >= 9.0 + 4.0  # +9 mapping on top of the 4-pixel border
    with pytest.raises(RuntimeError, match="shape of the mask"):
        model.event_extractor(_t(ev), _t(mask))
    # the matcher consumes the mapped keypoints like any others
    m = model.matcher(ef, imf)
    exp = oracle.mnn(oe["sparse_descriptors"][0], oi["sparse_descriptors"][0], want_la=False)
    assert np.array_equal(_np(m["matches0"][0])[0], exp["matches0"])


def test_torch_free_c_abi_host():
    """examples/c_abi_host: a C++/HIP program that links libeinx_hip.so and runs two extractors + MNN with hipMalloc'ed
    buffers only (no Python, no torch below or above the boundary)."""
    import subprocess
    from helpers import ROOT
    exe = os.path.join(ROOT, "examples", "c_abi_host")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    r = subprocess.run([exe, "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "C ABI host: OK" in r.stdout
    assert r.stdout.count("event keypoints") == 3
