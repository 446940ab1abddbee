"""GPU tests (-m gpu), component: nextrows.
SURVEY 8f and row H: events -> voxel grid + mask (events.hip), MR / MMA / VDD on the device (metrics.hip), the evaluation harnesses, the un-frozen Matcher branch.
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

from helpers import (GOLDEN, close_and_record, la_bound, metric_case, metric_inputs, rep_inputs, row_checksums, split, sub_dict,
                     synth, synth_raw_events, train_inputs)
from gpu_support import (DEV, E2E, EVENTS, FTOL, METRICS, MNN, REPS, TRAIN, _Z, _assert_feats_equal_oracle, _build, _inputs, _np,
                         _t, _unfrozen_matcher, pkg)

pytestmark = pytest.mark.gpu


def test_voxel_grid_and_events_mask(oracle):
    """Round 4: the voxel grid is DETERMINISTIC -- per voxel the contributions are added in the reference's serial order (corner
    major, then event order; representations.py:94-114), so the un-normalised grid is bit-equal to the oracle at every size, to
    the reference's fixtures at every size (round 5: generated with one torch thread, where torch adds serially; 60k events are
    compared through per-row checksums of the bit patterns), and two runs give the same bits."""
    from importlib import import_module
    from helpers import synth_raw_events
    rep = import_module(pkg.__name__ + ".datasets.representations")
    cases = [EVENTS.cases[n] for n in ("int_p01", "frac_pm1", "full", "full_frac")]
    for c in cases:
        ev = synth_raw_events(c)
        size = (c["bins"], c["H"], c["W"])
        keep = {k: v.copy() for k, v in ev.items()}
        grid = _np(rep.events_to_voxel_grid(ev, size, normalize=True))
        raw = _np(rep.events_to_voxel_grid(ev, size, normalize=False))
        assert all(np.array_equal(ev[k], keep[k]) for k in ev)  # the caller's dict is left alone
        name = c["name"]
        assert np.array_equal(raw, oracle.voxel_grid(ev, size, normalize=False)), name
        assert np.array_equal(raw, _np(rep.events_to_voxel_grid(ev, size, normalize=False))), "two runs differ"
        assert np.array_equal(grid, _np(rep.events_to_voxel_grid(ev, size, normalize=True))), "two runs differ"
        # normalisation: float64 statistics summed slab by slab here, voxel by voxel in the oracle -> equal up to the rounding
        # of the mean / std to fp32 (bit-equal unless a sum sits on a rounding boundary)
        close_and_record(f"events.{name}.grid vs oracle", grid, oracle.voxel_grid(ev, size, normalize=True), atol=1e-6)
        if f"{name}.grid" in EVENTS:
            assert np.array_equal(raw, EVENTS[f"{name}.raw"]), name  # the reference's own bits
            close_and_record(f"events.{name}.grid vs reference", grid, EVENTS[f"{name}.grid"], atol=2e-5, rtol=1e-5)
        else:  # 60k events: the reference's own bits too (fixture generated with one torch thread = serial adds), via row checksums
            from helpers import row_checksums
            assert np.array_equal(raw.reshape(-1)[::7], EVENTS[f"{name}.raw.stride7"]), name
            rs, rx = row_checksums(raw)
            assert np.array_equal(rs, EVENTS[f"{name}.raw.rowsum"]) and np.array_equal(rx, EVENTS[f"{name}.raw.rowxor"]), name
            close_and_record(f"events.{name}.grid vs reference", grid.reshape(-1)[::7], EVENTS[f"{name}.grid.stride7"], atol=2e-5, rtol=1e-5)
        mask = _np(rep.events_mask_batch([ev], (c["W"], c["H"])))[0, 0]
        exp = np.unpackbits(EVENTS[f"{name}.mask"])[:c["H"] * c["W"]].astype(bool).reshape(c["H"], c["W"])
        assert np.array_equal(mask, exp)  # integer counts: bit exact
    # batched call == per-sample calls, bit for bit (one sample is empty)
    empty = {k: v[:0] for k, v in synth_raw_events(EVENTS.cases["int_p01"]).items()}
    small = [synth_raw_events(EVENTS.cases["int_p01"]), empty, synth_raw_events(dict(EVENTS.cases["int_p01"], seed=99, n=1000))]
    gb = _np(rep.events_to_voxel_grid_batch(small, (5, 40, 48), normalize=False))
    for b, e in enumerate(small):
        assert np.array_equal(gb[b], oracle.voxel_grid(e, (5, 40, 48), normalize=False))


def test_voxel_grid_collisions_and_out_of_range_events(oracle):
    """The order-sensitive cases: thousands of fractional events on a handful of pixels (every 64-event batch collides, within
    and across corners), coordinates outside the sensor on every side (x, y in [-3, W+2]), unsorted timestamps, one hot pixel
    taking a third of all events, and a slab-crossing geometry other than 346x260 -- all bit-equal to the sequential oracle."""
    from importlib import import_module
    rep = import_module(pkg.__name__ + ".datasets.representations")
    cases = [dict(seed=3, n=20000, H=260, W=346, bins=5, box=4), dict(seed=4, n=50000, H=260, W=346, bins=5, box=400),
             dict(seed=5, n=9000, H=97, W=131, bins=3, box=30), dict(seed=6, n=70000, H=480, W=640, bins=5, box=700),
             # one image row per slab and more (slab, band) lists than the LDS histogram of the prep kernel holds (global counters)
             dict(seed=8, n=20000, H=300, W=640, bins=16, box=700)]
    for c in cases:
        n, H, W = c["n"], c["H"], c["W"]
        x = synth.uniform(c["seed"], (n,), -3.0, min(W + 2.0, c["box"]))
        y = synth.uniform(c["seed"] + 1, (n,), -3.0, min(H + 2.0, c["box"]))
        hot = synth.uniform01(c["seed"] + 2, (n,)) < np.float32(0.33)
        x = np.where(hot, np.float32(W // 3) + np.float32(0.25), x).astype(np.float32)
        y = np.where(hot, np.float32(H // 2) + np.float32(0.5), y).astype(np.float32)
        t = 1.5e9 + np.cumsum(synth.uniform01(c["seed"] + 3, (n,)).astype(np.float64) * 1e-4 + 1e-6)
        if c["seed"] == 5:  # unsorted timestamps between the first and the last event
            t[1:-1] = t[1:-1][np.argsort(synth.uniform01(77, (n - 2,)))]
        p = np.where(hot | (synth.uniform01(c["seed"] + 4, (n,)) < np.float32(0.5)), np.float32(1), np.float32(-1)).astype(np.float32)
        ev = {"x": x, "y": y, "t": t, "p": p}
        size = (c["bins"], H, W)
        raw = _np(rep.events_to_voxel_grid(ev, size, normalize=False))
        exp = oracle.voxel_grid(ev, size, normalize=False)
        assert np.array_equal(raw, exp), (c, int((raw != exp).sum()), float(np.abs(raw - exp).max()))
        assert np.abs(exp).max() > 100  # the hot pixel really accumulates thousands of contributions
        assert np.array_equal(raw, _np(rep.events_to_voxel_grid(ev, size, normalize=False)))


@pytest.mark.parametrize("name", list(METRICS.cases))
def test_metric_classes_vs_reference(oracle, name):
    """reference-named metric classes (update_one) -> metrics.hip -> the reference's own numbers: (y, x) and (x, y) rows, two
    image sizes, warps that push points off the image / leave nothing visible, an empty side, thresholds 1 / 3 / 5."""
    from importlib import import_module
    from helpers import metric_case, metric_inputs
    mm = import_module(pkg.__name__ + ".core.metrics.matching_metrics")
    km = import_module(pkg.__name__ + ".core.metrics.keypoints_metrics")
    c = METRICS.cases[name]
    mc = metric_case(c)
    k0, k1, d0, d1, mk0, mk1 = [_t(a) for a in metric_inputs(c)]
    Hm = torch.eye(3) if c["hom"] is None else torch.tensor(c["hom"], dtype=torch.float32).reshape(3, 3)
    vals = {}
    vals.update(mm.MatchingRatio("MR").update_one(mk0, mk1, k0, k1))
    for t in mc["thr"]:
        vals.update(mm.MeanMatchingAccuracy(f"MMA@{t}", threshold=t, ordering="xy" if mc["xy"] else "yx").update_one(mk0, mk1, Hm.to(DEV)))
    # ValidDescriptorsDistance's `ordering` names the opposite convention (keypoints_metrics.py:193-198), as in the generator
    vals.update(km.ValidDescriptorsDistance("VDD", mc["thr"], ordering="yx" if mc["xy"] else "xy").update_one(k0, k1, d0, d1, mc["size0"], mc["size1"],
                                                                                                       Hm.to(DEV)))
    names = ["MR"] + [f"MMA@{t}" for t in mc["thr"]] + [f"VDD_{p}@{t}" for t in mc["thr"] for p in ("Repeatability", "ValidDistance", "Angle")]
    got = np.array([vals[k] for k in names])
    exp = METRICS[f"{name}.values"]
    assert got.shape == exp.shape
    i = mc["idx"]
    np.testing.assert_allclose(got[i["counts"]], exp[i["counts"]], atol=1e-7, rtol=1e-6)
    np.testing.assert_allclose(got[i["dist"]], exp[i["dist"]], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(got[i["angle"]], exp[i["angle"]], atol=2e-3, rtol=1e-5)
    orc = oracle.pair_metrics(*metric_inputs(c), mc["size0"], mc["size1"], c["hom"], mma_thr=mc["thr"], vdd_thr=mc["thr"], kp_yx=not mc["xy"])
    np.testing.assert_allclose(got, orc, atol=1e-6, rtol=1e-6)


def test_batch_metrics_on_pipeline_output(oracle):
    """metrics of a whole EIM batch on the device == oracle metrics of each pair's outputs."""
    from importlib import import_module
    nm = import_module(pkg.__name__ + ".core.metrics._native_metrics")
    c = dict(E2E.cases["sp_mnn"])
    model, _ = _build(c, E2E)
    for ext in (model.event_extractor.extractor, model.image_extractor.extractor):
        ext.dense_outputs = False
    ev, mask, img = _inputs(c)
    evb, imb, mr = model.forward_batched(_t(ev), _t(img), _t(mask))
    out = _np(nm.batch_metrics(evb, imb, mr))
    ef, imf, m = model(_t(ev), _t(img), _t(mask))
    for b in range(c["B"]):
        exp = oracle.pair_metrics(_np(ef["sparse_positions"][b]), _np(imf["sparse_positions"][b]), _np(ef["sparse_descriptors"][b]),
                                  _np(imf["sparse_descriptors"][b]), _np(m["matched_kpts0"][b]), _np(m["matched_kpts1"][b]), (260, 346), (260, 346))
        np.testing.assert_allclose(out[b], exp, atol=1e-6, rtol=1e-6, equal_nan=True)


def test_same_time_harness_end_to_end(oracle):
    """row H: raw events -> voxel grid + mask -> EIM -> metrics, all on the device, against the
    oracle chain.  The voxel grid uses fp32 atomics (summation order), so the comparison is done on
    the harness' own voxel grid fed to the oracle extractors (bit-exact from there on)."""
    from helpers import synth_raw_events
    cfg = pkg.default_config("SP_MNN", event_channels=5)
    for sec in (cfg.event_extractor.vgg, cfg.image_extractor.superpointv1):
        sec.detection_top_k = 128
    model = pkg.EIM(cfg, device=DEV).eval()
    sdn = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=31)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()}, strict=False)
    for ext in (model.event_extractor.extractor, model.image_extractor.extractor):
        ext.dense_outputs = False
    H, W, B = 100, 124, 2
    evs = [synth_raw_events(dict(seed=300 + b, n=6000, H=H, W=W, bins=5, frac=False, pneg=False)) for b in range(B)]
    img = synth.synth_image(77, B, H, W)
    evalr = pkg.SameTimeEvaluator(model, bins=5, resolution=(W, H))
    rows, (ef, imf, m) = evalr.step(evs, _t(img))
    rows = _np(rows)
    # oracle chain from the same voxel grid / mask
    grid = _np(evalr.last_inputs[0])
    mask = np.stack([oracle.events_mask(e, (W, H)) for e in evs])[:, None]
    assert np.array_equal(_np(evalr.last_inputs[1]), mask)
    for b in range(B):
        np.testing.assert_allclose(grid[b], oracle.voxel_grid(evs[b], (5, H, W)), atol=2e-5, rtol=1e-5)
    oe = oracle.extractor_forward("vgg", sub_dict(sdn, "event_extractor.extractor."), grid.copy(), mask, top_k=128)
    oi = oracle.extractor_forward("superpointv1", sub_dict(sdn, "image_extractor.extractor."), img.copy(), None, top_k=128)
    _assert_feats_equal_oracle(ef, oe)
    _assert_feats_equal_oracle(imf, oi)
    for b in range(B):
        r = oracle.mnn(oe["sparse_descriptors"][b], oi["sparse_descriptors"][b], want_la=False)
        mk0, mk1 = oracle.matched_kpts(oe["sparse_positions"][b], oi["sparse_positions"][b], r["matches0"], 3)
        exp = oracle.pair_metrics(oe["sparse_positions"][b], oi["sparse_positions"][b], oe["sparse_descriptors"][b], oi["sparse_descriptors"][b],
                                  mk0, mk1, (H, W), (H, W))
        np.testing.assert_allclose(rows[b], exp, atol=1e-6, rtol=1e-6, equal_nan=True)
    res = evalr.result()
    assert set(res) == {"MR", "MMA@1", "MMA@3", "VDD_Repeatability@1", "VDD_ValidDistance@1", "VDD_Angle@1", "VDD_Repeatability@3",
                        "VDD_ValidDistance@3", "VDD_Angle@3"}


def test_different_time_harness_end_to_end(oracle):
    """test_events-image_different_time.py:187-264: events of frame i, image of a later frame j, related by a known
    (non-identity) homography per pair.  The evaluator's metric rows equal the oracle chain under the same homographies
    (the metric arithmetic itself is pinned to the reference's classes under a non-identity H by metrics.npz / r2.npz), and
    pose_inputs() hands over what the reference gives RelativePoseEstimation: the matched keypoint rows and their (x, y) views."""
    from helpers import synth_raw_events
    cfg = pkg.default_config("SP_MNN", event_channels=5)
    for sec in (cfg.event_extractor.vgg, cfg.image_extractor.superpointv1):
        sec.detection_top_k = 128
    model = pkg.EIM(cfg, device=DEV).eval()
    sdn = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=33)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()}, strict=False)
    for ext in (model.event_extractor.extractor, model.image_extractor.extractor):
        ext.dense_outputs = False
    H, W, B = 100, 124, 2
    evs = [synth_raw_events(dict(seed=400 + b, n=6000, H=H, W=W, bins=5, frac=False, pneg=False)) for b in range(B)]  # frames i
    img = synth.synth_image(91, B, H, W)                                                                                # frames j > i
    homs = np.array([[1.02, 0.015, -3.0, -0.01, 0.98, 2.5, 1e-5, -2e-5, 1.0],
                     [0.97, -0.02, 4.0, 0.03, 1.01, -1.5, -1e-5, 1e-5, 1.0]], np.float32).reshape(B, 3, 3)
    evalr = pkg.DifferentTimeEvaluator(model, bins=5, resolution=(W, H))
    rows, (ef, imf, m) = evalr.step(evs, _t(img), torch.from_numpy(homs).to(DEV))
    rows = _np(rows)
    grid = _np(evalr.last_inputs[0])
    mask = np.stack([oracle.events_mask(e, (W, H)) for e in evs])[:, None]
    oe = oracle.extractor_forward("vgg", sub_dict(sdn, "event_extractor.extractor."), grid.copy(), mask, top_k=128)
    oi = oracle.extractor_forward("superpointv1", sub_dict(sdn, "image_extractor.extractor."), img.copy(), None, top_k=128)
    _assert_feats_equal_oracle(ef, oe)
    _assert_feats_equal_oracle(imf, oi)
    ident = []
    for b in range(B):
        r = oracle.mnn(oe["sparse_descriptors"][b], oi["sparse_descriptors"][b], want_la=False)
        mk0, mk1 = oracle.matched_kpts(oe["sparse_positions"][b], oi["sparse_positions"][b], r["matches0"], 3)
        args = (oe["sparse_positions"][b], oi["sparse_positions"][b], oe["sparse_descriptors"][b], oi["sparse_descriptors"][b], mk0, mk1,
                (H, W), (H, W))
        np.testing.assert_allclose(rows[b], oracle.pair_metrics(*args, hom=homs[b]), atol=1e-6, rtol=1e-6, equal_nan=True)
        ident.append(oracle.pair_metrics(*args))
        p = evalr.pose_inputs(m, b)
        assert np.array_equal(_np(p["matched_kpts0"]), mk0) and np.array_equal(_np(p["matched_kpts1"]), mk1)  # [M,3] rows (y, x, score)
        assert model.event_extractor.extractor.ordering == "yx"
        assert np.array_equal(_np(p["matched_xy0"]), mk0[:, 1::-1]) and np.array_equal(_np(p["matched_xy1"]), mk1[:, 1::-1])
    # the homography really takes part: MR is motion-independent, the warped-distance metrics are not
    ident = np.array(ident)
    assert np.array_equal(rows[:, 0], ident[:, 0])
    assert not np.allclose(np.nan_to_num(rows[:, 1:]), np.nan_to_num(ident[:, 1:]))
    assert set(evalr.result()) == set(evalr.names)


@pytest.mark.parametrize("name", list(TRAIN.cases))
def test_unfrozen_matcher_vs_reference_golden(name):
    """the reference's padded + stacked inputs (fixture) through the native batched call: whole-batch
    tensors, per-pair matched keypoints, similarity (MNN) / all-layer ref_descriptors (LightGlue)."""
    c = TRAIN.cases[name]
    mm, _ = _unfrozen_matcher(name, c["L"])
    P0, D0, P1, D1 = (TRAIN[f"{name}.in_{k}"] for k in ("pos0", "desc0", "pos1", "desc1"))
    B = P0.shape[0]
    size = torch.tensor([260, 346])
    f0 = {"sparse_positions": [_t(P0[b]) for b in range(B)], "sparse_descriptors": [_t(D0[b]) for b in range(B)], "image_size": [size] * B}
    f1 = {"sparse_positions": [_t(P1[b]) for b in range(B)], "sparse_descriptors": [_t(D1[b]) for b in range(B)], "image_size": [size] * B}
    r = mm(f0, f1)
    assert r["input_feats0"] is f0 and tuple(f0["sparse_positions"].shape) == (B, c["L"], 3)  # stacked in place, like the reference
    assert np.array_equal(_np(r["matches0"]), TRAIN[f"{name}.matches0"]) and r["matches0"].dtype == torch.int64
    assert np.array_equal(_np(r["matches1"]), TRAIN[f"{name}.matches1"])
    np.testing.assert_allclose(_np(r["matching_scores0"]), TRAIN[f"{name}.matching_scores0"], atol=FTOL)
    np.testing.assert_allclose(_np(r["matching_scores1"]), TRAIN[f"{name}.matching_scores1"], atol=FTOL)
    np.testing.assert_allclose(_np(r["log_assignment"]), TRAIN[f"{name}.log_assignment"], atol=la_bound("lg.d256"), rtol=0)
    for b in range(B):
        np.testing.assert_allclose(_np(r["matched_kpts0"][b]), TRAIN[f"{name}.matched_kpts0.{b}"], atol=1e-6)
        np.testing.assert_allclose(_np(r["matched_kpts1"][b]), TRAIN[f"{name}.matched_kpts1.{b}"], atol=1e-6)
    if name == "mnn":
        np.testing.assert_allclose(_np(r["similarity"]), TRAIN[f"{name}.similarity"], atol=1e-6)
        assert "ref_descriptors0" not in r
    else:
        assert tuple(r["ref_descriptors0"].shape) == tuple(TRAIN[f"{name}.ref_shape"])  # [B, 9, L, 256]
        np.testing.assert_allclose(_np(r["ref_descriptors0"])[:, :, ::8, ::16], TRAIN[f"{name}.ref0_probe"], atol=FTOL, rtol=FTOL)
        np.testing.assert_allclose(_np(r["ref_descriptors1"])[:, :, ::8, ::16], TRAIN[f"{name}.ref1_probe"], atol=FTOL, rtol=FTOL)
        assert np.array_equal(_np(r["prune0"]), TRAIN[f"{name}.prune0"])
        # eval mode keeps only the last layer: [B,1,L,256], equal to the last slice of the training output
        mm.matcher.eval()
        f0e = {"sparse_positions": _t(P0), "sparse_descriptors": _t(D0), "image_size": [size] * B}
        f1e = {"sparse_positions": _t(P1), "sparse_descriptors": _t(D1), "image_size": [size] * B}
        re = mm.matcher(f0e, f1e)
        assert tuple(re["ref_descriptors0"].shape) == (B, 1, c["L"], 256)
        assert torch.equal(re["ref_descriptors0"][:, 0], r["ref_descriptors0"][:, -1])
        assert torch.equal(re["matches0"], r["matches0"])


@pytest.mark.parametrize("name", list(TRAIN.cases))
def test_unfrozen_matcher_random_padding(oracle, name):
    """ragged samples -> random padding to max_points_num.  The draws come from torch's generators
    (device generator for the positions, CPU generator for the descriptors) in the reference's call
    order, so re-seeding and re-drawing in the test predicts them; everything after the draws is
    checked bit for bit against the oracle."""
    from helpers import train_inputs
    c = TRAIN.cases[name]
    L = c["L"]
    mm, sd = _unfrozen_matcher(name, L)
    p0, d0, p1, d1 = train_inputs(c)
    B = len(p0)
    size = torch.tensor([260, 346], device=DEV)
    f0 = {"sparse_positions": [_t(a) for a in p0], "sparse_descriptors": [_t(a) for a in d0], "image_size": [size] * B}
    f1 = {"sparse_positions": [_t(a) for a in p1], "sparse_descriptors": [_t(a) for a in d1], "image_size": [size] * B}
    torch.manual_seed(c["tseed"])
    r = mm(f0, f1)
    torch.manual_seed(c["tseed"])
    exp = []
    for pos, desc in ((p0, d0), (p1, d1)):
        P, Dd = [], []
        for i in range(B):
            k = L - len(pos[i])
            u = torch.rand(k, 2, device=DEV).cpu().numpy() if k > 0 else None
            g = torch.randn(k, c["D"]).numpy() if k > 0 else None
            P.append(oracle.pad_positions(pos[i], L, u, (260, 346)))
            Dd.append(oracle.pad_descriptors(desc[i], L, g, 1.0))
        exp.append((np.stack(P), np.stack(Dd)))
    (P0, D0), (P1, D1) = exp
    assert np.array_equal(_np(r["input_feats0"]["sparse_positions"]), P0)
    assert np.array_equal(_np(r["input_feats1"]["sparse_positions"]), P1)
    assert np.array_equal(_np(r["input_feats0"]["sparse_descriptors"]), D0)
    assert np.array_equal(_np(r["input_feats1"]["sparse_descriptors"]), D1)
    pad = P0[0, c["counts0"][0]:]
    assert pad.shape[0] > 0 and np.all(pad[:, 2] == 0) and np.all(pad[:, 0] < 260) and np.all(pad[:, 1] < 346) and np.all(pad[:, :2] >= 0)
    if name == "mnn":
        o = oracle.mnn_stacked(P0, D0, P1, D1)
        for k in ("matches0", "matches1", "matching_scores0", "matching_scores1", "similarity"):
            assert np.array_equal(_np(r[k]), o[k]), k
        np.testing.assert_allclose(_np(r["log_assignment"]), o["log_assignment"], atol=1e-5)
        for b in range(B):
            assert np.array_equal(_np(r["matched_kpts0"][b]), o["matched_kpts0"][b])
            assert np.array_equal(_np(r["matched_kpts1"][b]), o["matched_kpts1"][b])
    else:
        o = oracle.lightglue_stacked(sd, P0, D0, P1, D1, (260, 346), (260, 346), training=True)
        assert np.array_equal(_np(r["matches0"]), o["matches0"]) and np.array_equal(_np(r["matches1"]), o["matches1"])
        np.testing.assert_allclose(_np(r["matching_scores0"]), o["matching_scores0"], atol=FTOL)
        np.testing.assert_allclose(_np(r["log_assignment"]), o["log_assignment"], atol=la_bound("lg.d256"), rtol=0)
        np.testing.assert_allclose(_np(r["ref_descriptors0"]), o["ref_descriptors0"], atol=FTOL, rtol=FTOL)
        np.testing.assert_allclose(_np(r["ref_descriptors1"]), o["ref_descriptors1"], atol=FTOL, rtol=FTOL)
        for b in range(B):
            assert np.array_equal(_np(r["matched_kpts0"][b]), o["matched_kpts0"][b])  # normalised coordinates
            assert np.array_equal(_np(r["matched_kpts1"][b]), o["matched_kpts1"][b])


def test_unfrozen_matcher_inside_eim_and_zero_pad_mode(oracle):
    """EIM.forward with matcher.freeze: false (EIM.py:92-95): extractor dicts are padded/stacked in
    place and handed to one batched matcher call; pad_mode 'zeros' is deterministic."""
    cfg = pkg.default_config("SP_MNN", event_channels=5)
    for sec in (cfg.event_extractor.vgg, cfg.image_extractor.superpointv1):
        sec.detection_top_k = 100
    cfg.matcher.freeze = False
    cfg.matcher.max_points_num = 128
    cfg.matcher.pad_mode = "zeros"
    model = pkg.EIM(cfg, device=DEV).eval()
    sdn = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=41)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()}, strict=False)
    for ext in (model.event_extractor.extractor, model.image_extractor.extractor):
        ext.dense_outputs = False
    B, H, W = 2, 120, 152
    ev, mask = synth.synth_events(62, B, 5, H, W)
    img = synth.synth_image(62, B, H, W)
    ef, imf, m = model(_t(ev), _t(img), _t(mask))
    assert m["input_feats0"] is ef and m["input_feats1"] is imf
    assert tuple(ef["sparse_positions"].shape) == (B, 128, 3) and tuple(imf["sparse_descriptors"].shape) == (B, 128, 256)
    oe = oracle.extractor_forward("vgg", sub_dict(sdn, "event_extractor.extractor."), ev.copy(), mask, top_k=100)
    oi = oracle.extractor_forward("superpointv1", sub_dict(sdn, "image_extractor.extractor."), img.copy(), None, top_k=100)
    P0 = np.stack([oracle.pad_positions(p, 128, None, (H, W), mode="zeros") for p in oe["sparse_positions"]])
    D0 = np.stack([oracle.pad_descriptors(d, 128, None, 1.0, mode="zeros") for d in oe["sparse_descriptors"]])
    P1 = np.stack([oracle.pad_positions(p, 128, None, (H, W), mode="zeros") for p in oi["sparse_positions"]])
    D1 = np.stack([oracle.pad_descriptors(d, 128, None, 1.0, mode="zeros") for d in oi["sparse_descriptors"]])
    assert np.array_equal(_np(ef["sparse_positions"]), P0) and np.array_equal(_np(ef["sparse_descriptors"]), D0)
    assert np.array_equal(_np(imf["sparse_positions"]), P1) and np.array_equal(_np(imf["sparse_descriptors"]), D1)
    o = oracle.mnn_stacked(P0, D0, P1, D1)
    assert tuple(m["matches0"].shape) == (B, 128) and tuple(m["similarity"].shape) == (B, 128, 128)
    # the all-zero padding rows tie everywhere (similarity 0): the first index wins on both sides
    for k in ("matches0", "matches1", "matching_scores0", "similarity"):
        assert np.array_equal(_np(m[k]), o[k]), k


def test_unfrozen_helpers_bit_exact(oracle):
    N = pkg.native
    x = synth.uniform(91, (37, 128), -1, 1)
    x[5] = 0  # zero row -> eps clamp
    assert np.array_equal(_np(N.normalize_rows(_t(x), 1.41)), oracle.normalize_rows(x, 1.41))
    k = np.concatenate([synth.uniform(92, (3, 50, 2), 0, 260), synth.uniform01(93, (3, 50, 1))], -1).astype(np.float32)
    got = _np(N.normalize_keypoints(_t(k), (260, 346)))
    assert got.shape == (3, 50, 2) and np.array_equal(got, oracle.normalize_keypoints(k, (260, 346)))
    got3 = _np(N.normalize_keypoints(_t(k), (260, 346), out_cols=3))
    assert np.array_equal(got3[..., :2], got) and np.all(got3[..., 2] == 0)
    u = synth.uniform01(94, (11, 2))
    assert np.array_equal(_np(N.random_positions(_t(u), (260, 346))), oracle.pad_positions(np.zeros((0, 3), np.float32), 11, u, (260, 346)))
    d0 = synth.synth_unit_descriptors(95, 70, 256, 1.0)
    d1 = synth.synth_unit_descriptors(96, 200, 256, 1.0)
    n = torch.tensor([70, 50], dtype=torch.int32, device=DEV)
    m = torch.tensor([200, 130], dtype=torch.int32, device=DEV)
    sim = _np(N.similarity(_t(np.stack([d0, d0])), n, _t(np.stack([d1, d1])), m))
    exp = oracle.mnn(d0, d1, want_la=False, want_sim=True)["similarity"]
    assert np.array_equal(sim[0], exp)
    assert np.array_equal(sim[1, :50, :130], exp[:50, :130]) and np.all(sim[1, 50:] == 0) and np.all(sim[1, :, 130:] == 0)


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


def test_voxel_grid_degenerate_time_stamps(oracle):
    """Found by tools/fuzz_parity.py: with all time stamps equal t_norm is NaN; `v_cvt_i32_f32` maps NaN to bin 0 (in range), torch's
    CPU `.int()` to INT_MIN (out of range) -- the kernel added NaN weights where the reference leaves the grid zero.  Now equal to the
    reference-generated fixture and to the oracle, single events and one-stamp bursts, batched with an ordinary sample."""
    from importlib import import_module
    from helpers import GOLDEN
    rep = import_module(pkg.__name__ + ".datasets.representations")
    z = np.load(os.path.join(GOLDEN, "events_degenerate.npz"))
    names = sorted({k.split(".")[0] for k in z.files if "." in k})
    for name in names:
        ev = {k: z[f"{name}.{k}"] for k in ("x", "y", "t", "p")}
        size = tuple(int(v) for v in z[f"{name}.size"])
        for norm in (False, True):
            got = _np(rep.events_to_voxel_grid({k: v.copy() for k, v in ev.items()}, size, normalize=norm))
            exp = z[f"{name}.grid_norm{int(norm)}"]
            if norm and name == "two_stamps":  # the reference's mean / std are torch reductions: 1e-5, as for the other fixtures
                np.testing.assert_allclose(got, exp, atol=1e-5, rtol=1e-5)
            else:
                assert np.array_equal(got, exp), (name, norm)
            assert np.array_equal(got, oracle.voxel_grid(ev, size, normalize=norm))
    # a degenerate sample next to an ordinary one in one batch
    a = {k: z[f"burst_one_stamp.{k}"] for k in ("x", "y", "t", "p")}
    b = {k: z[f"two_stamps.{k}"] for k in ("x", "y", "t", "p")}
    size = tuple(int(v) for v in z["two_stamps.size"])
    got = _np(rep.events_to_voxel_grid_batch([a, b, a], size, normalize=True))
    assert np.count_nonzero(got[0]) == 0 and np.count_nonzero(got[2]) == 0
    assert np.array_equal(got[1], oracle.voxel_grid(b, size, normalize=True))


def test_event_batches_without_any_event(oracle):
    """Found by tools/fuzz_parity.py --harness: a batch whose samples are ALL empty hands NULL event arrays to the C ABI, which
    refused them ("null pointer") although one empty sample among others had always given a zero grid.  Zero grids and all-false
    masks, also through the evaluator step."""
    from importlib import import_module
    rep = import_module(pkg.__name__ + ".datasets.representations")
    empty = {"x": np.zeros(0, np.float32), "y": np.zeros(0, np.float32), "t": np.zeros(0, np.float64), "p": np.zeros(0, np.float32)}
    grid = _np(rep.events_to_voxel_grid_batch([empty, empty], (5, 40, 56)))
    mask = _np(rep.events_mask_batch([empty, empty], (56, 40)))
    assert grid.shape == (2, 5, 40, 56) and not grid.any()
    assert mask.shape == (2, 1, 40, 56) and not mask.any()
    one = {"x": np.array([3.5], np.float32), "y": np.array([2.25], np.float32), "t": np.array([1.0]), "p": np.array([1.0], np.float32)}
    g2 = _np(rep.events_to_voxel_grid_batch([empty, one], (5, 40, 56)))
    assert not g2.any()  # (one event: NaN t_norm, dropped like in the reference)
    m2 = _np(rep.events_mask_batch([empty, one], (56, 40)))
    assert not m2[0].any() and m2[1].sum() == 1 and m2[1, 0, 2, 3]


def test_harness_run_streams_batches_with_the_results_of_step():
    """SameTimeEvaluator.run (events packed into page-locked memory, uploaded on a side stream and enqueued while the
    previous batch is still on the device) yields, batch by batch, exactly what step() returns: metric rows, keypoints,
    descriptors and matches bit for bit -- over batches of different event counts, integer-typed event arrays, a batch
    without any event and per-pair homographies; the accumulated means are equal too."""
    from helpers import synth, synth_raw_events
    cfg = pkg.default_config("SP_MNN", event_channels=5)
    for sec in (cfg.event_extractor.vgg, cfg.image_extractor.superpointv1):
        sec.detection_top_k = 128
    model = pkg.EIM(cfg, device=DEV).eval()
    sdn = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=35)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()}, strict=False)
    H, W, B = 100, 124, 3
    batches = []
    for i, n in enumerate((6000, 900, 0, 12000, 3000)):
        evs = [synth_raw_events(dict(seed=600 + 10 * i + b, n=n + 37 * b if n else 0, H=H, W=W, bins=5, frac=False, pneg=False)) for b in range(B)]
        if i == 1:  # what an HDF5 loader hands over: integer coordinates / polarities
            evs = [dict(x=e["x"].astype(np.int16), y=e["y"].astype(np.int16), t=e["t"], p=e["p"].astype(np.int8)) for e in evs]
        hom = None
        if i % 2:
            hom = _t(np.tile(np.array([[1.01, 0.01, -2.0], [-0.01, 0.99, 1.5], [1e-5, -1e-5, 1.0]], np.float32), (B, 1, 1)))
        batches.append((evs, synth.synth_image(80 + i, B, H, W), hom))
    a = pkg.DifferentTimeEvaluator(model, bins=5, resolution=(W, H))
    exp = [a.step(evs, _t(img), hom) for evs, img, hom in batches]
    b = pkg.DifferentTimeEvaluator(model, bins=5, resolution=(W, H))
    for depth in (2, 3, 1):
        got = list(b.run(((evs, _t(img), hom) for evs, img, hom in batches), depth=depth))
        assert len(got) == len(exp)
        for (r0, (e0, i0, m0)), (r1, (e1, i1, m1)) in zip(exp, got):
            assert torch.equal(torch.nan_to_num(r0, nan=-7.0), torch.nan_to_num(r1, nan=-7.0))
            for f0, f1 in ((e0, e1), (i0, i1)):
                for key in ("sparse_positions", "sparse_descriptors"):
                    assert all(torch.equal(x, y) for x, y in zip(f0[key], f1[key]))
            assert all(torch.equal(x, y) for x, y in zip(m0["matches0"], m1["matches0"]))
            assert all(torch.equal(x, y) for x, y in zip(m0["matched_kpts1"], m1["matched_kpts1"]))
    ra, rb = a.result(), b.result()  # b saw every batch three times: same means
    for k in ra:
        assert (ra[k] != ra[k] and rb[k] != rb[k]) or abs(ra[k] - rb[k]) <= 1e-12 * max(1.0, abs(ra[k])), k
