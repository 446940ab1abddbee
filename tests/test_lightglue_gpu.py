"""GPU tests (-m gpu), component: lightglue.
SURVEY 8a row A21 (LightGlue: nine layers, other widths / head counts, input_proj, weight folding, small-grid kernels): lightglue.hip.
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

from helpers import (Golden, close_and_record, la_bound, lg_inputs, record_flips, synth)
from gpu_support import (BATCH_CASES, DEV, FTOL, LG, LGCAL, LGCFG, _lg_model, _lgcfg_model, _np, _t, pkg)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["d256", "d128", "full"])
def test_lightglue_vs_golden(oracle, name):
    from helpers import lg_inputs
    c = LG.cases[name]
    lg, sd = _lg_model(c)
    d0, d1, k0, k1 = lg_inputs(c)
    size = torch.tensor([260, 346])
    f0 = {"sparse_descriptors": _t(d0)[None], "sparse_positions": _t(k0)[None], "image_size": [size]}
    f1 = {"sparse_descriptors": _t(d1)[None], "sparse_positions": _t(k1)[None], "image_size": [size]}
    r = lg(f0, f1)
    # bit-exact match assignments against the reference
    assert record_flips(f"lg.{name}.matches0 vs reference", _np(r["matches0"]), LG[f"{name}.matches0"]) == 0
    assert record_flips(f"lg.{name}.matches1 vs reference", _np(r["matches1"]), LG[f"{name}.matches1"]) == 0
    close_and_record(f"lg.{name}.matching_scores0 vs reference", _np(r["matching_scores0"]), LG[f"{name}.mscores0"], atol=FTOL)
    close_and_record(f"lg.{name}.matching_scores1 vs reference", _np(r["matching_scores1"]), LG[f"{name}.mscores1"], atol=FTOL)
    assert np.array_equal(_np(r["matched_kpts0"]), LG[f"{name}.matched_kpts0"])
    assert np.array_equal(_np(r["matched_kpts1"]), LG[f"{name}.matched_kpts1"])
    la = _np(r["log_assignment"])
    assert la.shape == (1, c["n"] + 1, c["m"] + 1)
    bound = la_bound(f"lg.{name}")
    if f"{name}.la" in LG:
        close_and_record(f"lg.{name}.log_assignment vs reference", la, LG[f"{name}.la"], atol=bound)
        close_and_record(f"lg.{name}.log_assignment vs reference in float64", la[0], LGCAL[f"lg.{name}.la_f64"], atol=bound)
    else:
        close_and_record(f"lg.{name}.log_assignment vs reference", la[0, ::37, ::41], LG[f"{name}.la_probe"], atol=bound)
        close_and_record(f"lg.{name}.log_assignment vs reference in float64", la[0, ::37, ::41], LGCAL[f"lg.{name}.la_f64"], atol=bound)
    sn = max(1, c["n"] // 16)
    ref = _np(r["ref_descriptors0"])
    assert ref.shape == (1, 1, c["n"], 256)
    close_and_record(f"lg.{name}.ref_descriptors0 vs reference", ref[0, 0, ::sn, ::16], LG[f"{name}.ref_desc0_probe"], atol=FTOL)
    assert tuple(r["prune0"].shape) == (1, c["n"]) and float(r["prune0"][0, 0]) == 9.0
    if name != "full":
        exp = oracle.lightglue(sd, k0, d0, k1, d1)
        assert record_flips(f"lg.{name}.matches0 vs oracle", _np(r["matches0"])[0], exp["matches0"], exp["log_assignment"]) == 0
        close_and_record(f"lg.{name}.log_assignment vs oracle", la[0], exp["log_assignment"], atol=bound)
        close_and_record(f"lg.{name}.ref_descriptors0 vs oracle", ref[0, 0], exp["ref_descriptors0"], atol=FTOL)


def test_lightglue_weight_folding_is_equivalent():
    """fold_message_projection (out_proj / to_out folded into the FFN's first Linear at load time)
    must not change the assignment and may move floats only at rounding level."""
    from helpers import lg_inputs
    c = LG.cases["d256"]
    lg, _ = _lg_model(c)
    d0, d1, k0, k1 = lg_inputs(c)
    size = torch.tensor([260, 346])
    f0 = {"sparse_descriptors": _t(d0)[None], "sparse_positions": _t(k0)[None], "image_size": [size]}
    f1 = {"sparse_descriptors": _t(d1)[None], "sparse_positions": _t(k1)[None], "image_size": [size]}
    lg.fold_message_projection = True
    lg.refresh()
    a = lg(f0, f1)
    lg.fold_message_projection = False
    lg.refresh()
    b = lg(f0, f1)
    assert torch.equal(a["matches0"], b["matches0"]) and torch.equal(a["matches1"], b["matches1"])
    np.testing.assert_allclose(_np(a["log_assignment"]), _np(b["log_assignment"]), atol=la_bound("lg.d256"), rtol=0)
    np.testing.assert_allclose(_np(a["ref_descriptors0"]), _np(b["ref_descriptors0"]), atol=2e-5, rtol=1e-5)


# ------------------------------------------------------------------ LightGlue: both sides stacked in one launch == one launch per side
@pytest.mark.parametrize("B,n,m", [(1, 1024, 1024), (3, 300, 300), (2, 517, 480)])
def test_lightglue_stacked_sides_equal_the_per_side_path(B, n, m):
    """Equal capacities run every layer once over 2B entries (cross attention reads the partner entry); a side-1 batch
    padded by one unused row has another capacity and takes the launch-per-side path: every output must be bit-identical."""
    from importlib import import_module
    N = pkg.native
    PairBatch = import_module(pkg.__name__ + ".core.modules.matchers._batched").PairBatch
    LG = import_module(pkg.__name__ + ".core.modules.matchers.lightglue").LightGlue
    lg = LG({"input_dim": 256}).to(DEV).eval()
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in lg.state_dict().items()], seed=77)
    lg.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    cap = max(n, m)
    rng = np.random.default_rng(B * 1000 + n)

    def side(cnt, cap_):
        pb = PairBatch()
        k = np.zeros((B, cap_, 3), np.float32)
        d = np.zeros((B, cap_, 256), np.float32)
        k[:, :cnt, 0] = rng.uniform(0, 260, (B, cnt))
        k[:, :cnt, 1] = rng.uniform(0, 346, (B, cnt))
        v = rng.standard_normal((B, cnt, 256)).astype(np.float32)
        d[:, :cnt] = v / np.linalg.norm(v, axis=-1, keepdims=True)
        pb.kpts, pb.desc = _t(k), _t(d)
        pb.counts = torch.tensor([cnt] * (B - 1) + [max(cnt - 7, 1)], dtype=torch.int32, device=DEV)  # one ragged entry
        pb.cap, pb.B, pb.image_size, pb.counts_host = cap_, B, (260, 346), None
        return pb

    pb0, pb1 = side(n, cap), side(m, cap)
    w = lg._pack()[0]
    a = N.lightglue(w, pb0, pb1, want_la=True, want_ref=True)
    pb1p = PairBatch()
    pb1p.kpts = torch.cat([pb1.kpts, torch.zeros(B, 1, 3, device=DEV)], 1).contiguous()
    pb1p.desc = torch.cat([pb1.desc, torch.zeros(B, 1, 256, device=DEV)], 1).contiguous()
    pb1p.counts, pb1p.cap, pb1p.B, pb1p.image_size, pb1p.counts_host = pb1.counts, cap + 1, B, (260, 346), None
    b = N.lightglue(w, pb0, pb1p, want_la=True, want_ref=True)
    cnt0, cnt1 = pb0.counts.tolist(), pb1.counts.tolist()
    for i in range(B):  # rows past an entry's count are never written
        c0, c1 = cnt0[i], cnt1[i]
        assert torch.equal(a.matches0[i, :c0], b.matches0[i, :c0]) and torch.equal(a.scores0[i, :c0], b.scores0[i, :c0])
        assert torch.equal(a.matches1[i, :c1], b.matches1[i, :c1]) and torch.equal(a.scores1[i, :c1], b.scores1[i, :c1])
        assert torch.equal(a.ref0[i, :c0], b.ref0[i, :c0]) and torch.equal(a.ref1[i, :c1], b.ref1[i, :c1])
        assert torch.equal(a.la[i, :c0, :c1], b.la[i, :c0, :c1])
    assert int((a.matches0 > -1).sum()) > 0


@pytest.mark.parametrize("n,m", [(1024, 1024), (700, 613)])
def test_lightglue_small_grid_kernels_equal_the_large_grid_kernels(n, m):
    """A single pair runs its linears on lg_gemm_small_kernel (64x64 tiles, fewer than 256 128x128 tiles) and its attention on
    lg_attn16_kernel (four waves share every key block of 16 / 32 queries on the 16x16x4 instruction), the same pair as
    entry 0 of a batch of 8 on lg_gemm_kernel / lg_attn_kernel: every output of the pair must be bit-identical (one k-ordered
    chain per output in both linears; the same chain of matrix steps, maxima, exponentials and sums in both attentions)."""
    from importlib import import_module
    N = pkg.native
    PairBatch = import_module(pkg.__name__ + ".core.modules.matchers._batched").PairBatch
    LG = import_module(pkg.__name__ + ".core.modules.matchers.lightglue").LightGlue
    lg = LG({"input_dim": 256}).to(DEV).eval()
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in lg.state_dict().items()], seed=78)
    lg.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    w = lg._pack()[0]
    cap = max(n, m)
    rng = np.random.default_rng(n)
    B = 8

    def arrays(cnt):
        k = np.zeros((B, cap, 3), np.float32)
        d = np.zeros((B, cap, 256), np.float32)
        k[:, :cnt, 0] = rng.uniform(0, 260, (B, cnt))
        k[:, :cnt, 1] = rng.uniform(0, 346, (B, cnt))
        v = rng.standard_normal((B, cnt, 256)).astype(np.float32)
        d[:, :cnt] = v / np.linalg.norm(v, axis=-1, keepdims=True)
        return k, d

    def batch(k, d, cnt, nb):
        pb = PairBatch()
        pb.kpts, pb.desc = _t(k[:nb]), _t(d[:nb])
        pb.counts = torch.full((nb,), cnt, dtype=torch.int32, device=DEV)
        pb.cap, pb.B, pb.image_size, pb.counts_host = cap, nb, (260, 346), None
        return pb

    (k0, d0), (k1, d1) = arrays(n), arrays(m)
    big = N.lightglue(w, batch(k0, d0, n, B), batch(k1, d1, m, B), want_la=True, want_ref=True)
    one = N.lightglue(w, batch(k0, d0, n, 1), batch(k1, d1, m, 1), want_la=True, want_ref=True)
    assert torch.equal(one.matches0[0, :n], big.matches0[0, :n]) and torch.equal(one.scores0[0, :n], big.scores0[0, :n])
    assert torch.equal(one.matches1[0, :m], big.matches1[0, :m]) and torch.equal(one.scores1[0, :m], big.scores1[0, :m])
    assert torch.equal(one.ref0[0, :n], big.ref0[0, :n]) and torch.equal(one.ref1[0, :m], big.ref1[0, :m])
    assert torch.equal(one.la[0, :n, :m], big.la[0, :n, :m])
    assert int((one.matches0[0, :n] > -1).sum()) > 0
    # two pairs: the attention's latency form with 32 queries per workgroup (round 5: lg_attn16_kernel<32>; one pair runs <16>)
    two = N.lightglue(w, batch(k0, d0, n, 2), batch(k1, d1, m, 2), want_la=True, want_ref=True)
    for i in range(2):
        assert torch.equal(two.matches0[i, :n], big.matches0[i, :n]) and torch.equal(two.scores0[i, :n], big.scores0[i, :n])
        assert torch.equal(two.ref0[i, :n], big.ref0[i, :n]) and torch.equal(two.ref1[i, :m], big.ref1[i, :m])
        assert torch.equal(two.la[i, :n, :m], big.la[i, :n, :m])


@pytest.mark.parametrize("B", [1, 8])
def test_lightglue_merged_qk_v_projection_equals_two_launches(B):
    """Round 5: CrossBlock.to_qk and to_v as ONE launch over the merged weight image [Wqk; Wv] (qk | v side by side in the FFN's
    hidden buffer, the attention reads them at row stride 512) against the two separate launches: bit-identical, on the
    single-pair kernels (lg_gemm_small_kernel / lg_attn16_kernel<512>) and on the batch kernels (lg_gemm_kernel / lg_attn_kernel<64,256,512>)."""
    from importlib import import_module
    PairBatch = import_module(pkg.__name__ + ".core.modules.matchers._batched").PairBatch
    LG = import_module(pkg.__name__ + ".core.modules.matchers.lightglue").LightGlue
    lg = LG({"input_dim": 256}).to(DEV).eval()
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in lg.state_dict().items()], seed=79)
    lg.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    n, m, cap = 1000, 1024, 1024
    rng = np.random.default_rng(B)

    def side(cnt):
        pb = PairBatch()
        k = np.zeros((B, cap, 3), np.float32)
        d = np.zeros((B, cap, 256), np.float32)
        k[:, :cnt, 0] = rng.uniform(0, 260, (B, cnt))
        k[:, :cnt, 1] = rng.uniform(0, 346, (B, cnt))
        v = rng.standard_normal((B, cnt, 256)).astype(np.float32)
        d[:, :cnt] = v / np.linalg.norm(v, axis=-1, keepdims=True)
        pb.kpts, pb.desc = _t(k), _t(d)
        pb.counts = torch.full((B,), cnt, dtype=torch.int32, device=DEV)
        pb.cap, pb.B, pb.image_size, pb.counts_host = cap, B, (260, 346), None
        return pb

    pb0, pb1 = side(n), side(m)
    outs = []
    for merged in (True, False):
        lg.merge_qk_v = merged
        lg.refresh()
        w = lg._pack()[0]
        assert bool(w.layers[0].Wqk_v) == merged
        outs.append(pkg.native.lightglue(w, pb0, pb1, want_la=True, want_ref=True))
    a, b = outs
    assert torch.equal(a.matches0[:, :n], b.matches0[:, :n]) and torch.equal(a.scores0[:, :n], b.scores0[:, :n])
    assert torch.equal(a.ref0[:, :n], b.ref0[:, :n]) and torch.equal(a.ref1[:, :m], b.ref1[:, :m])
    assert torch.equal(a.la[:, :n, :m], b.la[:, :n, :m])
    assert int((a.matches0[:, :n] > -1).sum()) > 0


def test_lightglue_filter_threshold_assigned_between_forwards(oracle):
    """The reference passes `self.conf.filter_threshold` to filter_matches in every forward (lightglue.py:656); the native weight image
    used to keep the value it was packed with.  Assigning it between two forwards changes the next one (eager and graph mode)."""
    from helpers import synth
    lg = pkg.LightGlue({"input_dim": 256}).to(DEV).eval()
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in lg.state_dict().items()], seed=59)
    lg.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    r = np.random.default_rng(59)
    n0, n1 = 120, 150
    d0 = r.uniform(-1, 1, (n0, 256)).astype(np.float32)
    d1 = r.uniform(-1, 1, (n1, 256)).astype(np.float32)
    d1[:60] = d0[:60] + r.uniform(-0.05, 0.05, (60, 256)).astype(np.float32)
    d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
    k0 = np.stack([r.uniform(0, 260, n0), r.uniform(0, 346, n0), r.uniform(0, 1, n0)], 1).astype(np.float32)
    k1 = np.stack([r.uniform(0, 260, n1), r.uniform(0, 346, n1), r.uniform(0, 1, n1)], 1).astype(np.float32)
    size = torch.tensor([260, 346])
    f0 = {"sparse_descriptors": _t(d0)[None], "sparse_positions": _t(k0)[None], "image_size": [size]}
    f1 = {"sparse_descriptors": _t(d1)[None], "sparse_positions": _t(k1)[None], "image_size": [size]}
    base = oracle.lightglue(sd, k0, d0, k1, d1, filter_threshold=0.0)
    scores = np.sort(base["matching_scores0"][np.asarray(base["matches0"]) > -1])
    assert len(scores) >= 4
    thr = float((scores[len(scores) // 2 - 1] + scores[len(scores) // 2]) / 2)  # a threshold between two matched scores
    counts = []
    for value in (0.0, thr, 0.0):
        lg.conf.filter_threshold = value
        got = lg(f0, f1)
        exp = oracle.lightglue(sd, k0, d0, k1, d1, filter_threshold=value)
        assert np.array_equal(_np(got["matches0"])[0], exp["matches0"]), value
        counts.append(int((np.asarray(exp["matches0"]) > -1).sum()))
    assert counts[1] < counts[0] == counts[2]


@pytest.mark.parametrize("fold", [True, False])
@pytest.mark.parametrize("name", list(LGCFG.cases))
def test_lightglue_other_widths_vs_reference_and_oracle(oracle, name, fold):
    c = LGCFG.cases[name]
    keys = json.loads(bytes(LGCFG[f"{name}.state_keys"]).decode())
    lg, sd = _lgcfg_model(c, keys)  # strict load: the parameter tree has the reference's names and shapes
    lg.fold_message_projection = fold
    d0, d1, k0, k1 = lg_inputs(c)
    size = torch.tensor([260, 346])
    f0 = {"sparse_descriptors": _t(d0)[None], "sparse_positions": _t(k0)[None], "image_size": [size]}
    f1 = {"sparse_descriptors": _t(d1)[None], "sparse_positions": _t(k1)[None], "image_size": [size]}
    r = lg(f0, f1)
    tag = f"lgcfg.{name}" + ("" if fold else ".unfolded")
    assert record_flips(f"{tag}.matches0 vs reference", _np(r["matches0"]), LGCFG[f"{name}.matches0"]) == 0
    assert record_flips(f"{tag}.matches1 vs reference", _np(r["matches1"]), LGCFG[f"{name}.matches1"]) == 0
    close_and_record(f"{tag}.matching_scores0 vs reference", _np(r["matching_scores0"]), LGCFG[f"{name}.mscores0"], atol=FTOL)
    close_and_record(f"{tag}.matching_scores1 vs reference", _np(r["matching_scores1"]), LGCFG[f"{name}.mscores1"], atol=FTOL)
    assert np.array_equal(_np(r["matched_kpts0"]), LGCFG[f"{name}.matched_kpts0"])
    assert np.array_equal(_np(r["matched_kpts1"]), LGCFG[f"{name}.matched_kpts1"])
    la = _np(r["log_assignment"])
    bound = la_bound(f"lgcfg.{name}")
    close_and_record(f"{tag}.log_assignment vs reference", la, LGCFG[f"{name}.la"], atol=bound)
    sn = max(1, c["n"] // 16)
    ref = _np(r["ref_descriptors0"])
    assert ref.shape == (1, 1, c["n"], c["descriptor_dim"])
    close_and_record(f"{tag}.ref_descriptors0 vs reference", ref[0, 0, ::sn, ::8], LGCFG[f"{name}.ref_desc0_probe"], atol=FTOL)
    assert np.array_equal(_np(r["prune0"]), LGCFG[f"{name}.prune0"])  # ones * n_layers
    exp = oracle.lightglue(sd, k0, d0, k1, d1, n_layers=c["n_layers"], heads=c["num_heads"])
    assert record_flips(f"{tag}.matches0 vs oracle", _np(r["matches0"])[0], exp["matches0"], exp["log_assignment"]) == 0
    close_and_record(f"{tag}.log_assignment vs oracle", la[0], exp["log_assignment"], atol=bound)
    close_and_record(f"{tag}.ref_descriptors0 vs oracle", ref[0, 0], exp["ref_descriptors0"], atol=FTOL)


@pytest.mark.parametrize("case", range(len(BATCH_CASES)))
def test_lightglue_other_widths_batched_vs_per_pair_oracle(oracle, case):
    """Ragged batches through the device-count path (both kernels of the linears, stacked and unstacked sides): every pair
    equals its own oracle run; padding rows hold garbage and stay unmatched."""
    conf, B, cap0, cap1 = BATCH_CASES[case]
    bt = import_module(pkg.__name__ + ".core.modules.matchers._batched")
    c = dict(conf, wseed=700 + case)
    lg, sd = _lgcfg_model(c)
    r = np.random.default_rng(4000 + case)
    din = conf["input_dim"]
    n0 = [int(r.integers(cap0 // 2, cap0 + 1)) for _ in range(B)]
    n1 = [int(r.integers(cap1 // 2, cap1 + 1)) for _ in range(B)]
    n0[0], n1[-1] = cap0, cap1
    n0[-1] = max(1, cap0 // 7)
    H, W = 260, 346
    K0, K1 = np.full((B, cap0, 3), 1e6, np.float32), np.zeros((B, cap1, 3), np.float32)
    D0, D1 = np.full((B, cap0, din), 7.0, np.float32), np.full((B, cap1, din), -3.0, np.float32)
    for b in range(B):
        d0 = r.uniform(-1, 1, (n0[b], din)).astype(np.float32)
        d1 = r.uniform(-1, 1, (n1[b], din)).astype(np.float32)
        k = min(n0[b], n1[b]) // 2
        d1[:k] = d0[:k] + r.uniform(-0.1, 0.1, (k, din)).astype(np.float32)
        D0[b, :n0[b]] = d0 / np.linalg.norm(d0, axis=1, keepdims=True)
        D1[b, :n1[b]] = d1 / np.linalg.norm(d1, axis=1, keepdims=True)
        K0[b, :n0[b]] = np.stack([r.uniform(0, H, n0[b]), r.uniform(0, W, n0[b]), r.uniform(0, 1, n0[b])], 1)
        K1[b, :n1[b]] = np.stack([r.uniform(0, H, n1[b]), r.uniform(0, W, n1[b]), r.uniform(0, 1, n1[b])], 1)
    pbs = []
    for K, D, cnt, cap in ((K0, D0, n0, cap0), (K1, D1, n1, cap1)):
        pb = bt.PairBatch()
        pb.kpts, pb.desc, pb.counts = _t(K), _t(D), _t(np.asarray(cnt, np.int32))
        pb.cap, pb.B, pb.image_size, pb.counts_host = cap, B, (H, W), None
        pbs.append(pb)
    mr = lg.match_batched(pbs[0], pbs[1])
    m0, s0, ref0 = _np(mr.matches0), _np(mr.scores0), _np(mr.ref0)
    tag = f"lgcfg.batch{case}"
    for b in (0, B - 1):  # the full-capacity pair and the short one (the oracle needs seconds per 1024-keypoint pair)
        exp = oracle.lightglue(sd, K0[b, :n0[b]], D0[b, :n0[b]], K1[b, :n1[b]], D1[b, :n1[b]], size0=(H, W), size1=(H, W),
                               n_layers=conf["n_layers"], heads=conf["num_heads"])
        close_and_record(f"{tag}.matching_scores0 vs oracle", s0[b, :n0[b]], exp["matching_scores0"], atol=FTOL)
        close_and_record(f"{tag}.ref_descriptors0 vs oracle", ref0[b, :n0[b]], exp["ref_descriptors0"], atol=FTOL)
        e0 = np.asarray(exp["matches0"])
        for i in np.nonzero(m0[b, :n0[b]] != e0)[0]:  # only arg-max near-ties may differ
            row = exp["log_assignment"][i, :-1]
            cand = [j for j in (m0[b, i], e0[i]) if j >= 0]
            near_tie = len(cand) == 2 and abs(row[cand[0]] - row[cand[1]]) < 2e-3
            near_thr = min(abs(float(np.exp(row[j]))) for j in cand) < 2e-3 if cand else False
            assert near_tie or near_thr, (tag, b, int(i), int(m0[b, i]), int(e0[i]))
        record_flips(f"{tag}.matches0 vs oracle", m0[b, :n0[b]], e0, exp["log_assignment"])
    for b in range(B):
        assert (m0[b, n0[b]:] == -1).all() and (m0[b, :n0[b]] < n1[b]).all()


def test_batched_equals_single_pairs_for_other_widths():
    """The stacked batch path and the single-pair path (64x64-tile linears) are the same arithmetic: equal bits per pair."""
    conf = dict(input_dim=128, descriptor_dim=128, num_heads=4, n_layers=3)
    lg, sd = _lgcfg_model(dict(conf, wseed=731))
    size = torch.tensor([260, 346])
    outs = []
    feats = []
    for s in range(3):
        d0, d1, k0, k1 = lg_inputs(dict(seed=500 + 10 * s, n=256, m=256, input_dim=128, shared=100))
        feats.append((d0, d1, k0, k1))
        f0 = {"sparse_descriptors": _t(d0)[None], "sparse_positions": _t(k0)[None], "image_size": [size]}
        f1 = {"sparse_descriptors": _t(d1)[None], "sparse_positions": _t(k1)[None], "image_size": [size]}
        outs.append(lg(f0, f1))
    f0 = {"sparse_descriptors": _t(np.stack([f[0] for f in feats])), "sparse_positions": _t(np.stack([f[2] for f in feats])), "image_size": [size] * 3}
    f1 = {"sparse_descriptors": _t(np.stack([f[1] for f in feats])), "sparse_positions": _t(np.stack([f[3] for f in feats])), "image_size": [size] * 3}
    rb = lg(f0, f1)
    for s in range(3):
        assert torch.equal(rb["matches0"][s], outs[s]["matches0"][0])
        assert torch.equal(rb["matching_scores0"][s], outs[s]["matching_scores0"][0])
        assert torch.equal(rb["log_assignment"][s], outs[s]["log_assignment"][0])
        assert torch.equal(rb["ref_descriptors0"][s], outs[s]["ref_descriptors0"][0])


def test_add_scale_ori_fails_like_the_reference():
    """add_scale_ori=True: the reference builds posenc.Wr as Linear(4, head_dim/2) (:457-459) and never appends scales /
    orientations (:540-560 commented out), so its forward raises in posenc; recorded from the reference in lgcfg.npz."""
    rec = LGCFG.meta["add_scale_ori"]
    lg = pkg.LightGlue({"input_dim": 256, "add_scale_ori": True}).to(DEV).eval()
    assert {k: list(v.shape) for k, v in lg.state_dict().items() if k.startswith("posenc")} == rec["state_keys"]
    c = Golden("lg").cases["d256"]
    d0, d1, k0, k1 = lg_inputs(c)
    size = torch.tensor([260, 346])
    f0 = {"sparse_descriptors": _t(d0)[None], "sparse_positions": _t(k0)[None], "image_size": [size]}
    f1 = {"sparse_descriptors": _t(d1)[None], "sparse_positions": _t(k1)[None], "image_size": [size]}
    assert rec["raises"] == "RuntimeError"
    with pytest.raises(RuntimeError) as e:
        lg(f0, f1)
    assert str(e.value) == rec["message"]


def test_unsupported_head_width_is_refused_loudly():
    for conf in ({"input_dim": 60, "descriptor_dim": 60, "num_heads": 2}, {"input_dim": 1024, "descriptor_dim": 1024, "num_heads": 2}):
        with pytest.raises(NotImplementedError):  # head widths that are not a multiple of 4 / wider than 256 (16-wide heads run since round 6)
            pkg.LightGlue(conf)
    with pytest.raises(AssertionError):
        pkg.LightGlue({"input_dim": 256, "descriptor_dim": 256, "num_heads": 3})  # the reference's own assert (:247)
    L = import_module(pkg.__name__ + "._native").lib()
    assert L.einx_lg_ws_bytes_heads(1, 64, 64, 60, 2, 60) == 0 and L.einx_lg_ws_bytes_heads(1, 64, 64, 1024, 2, 1024) == 0
    assert L.einx_lg_ws_bytes_heads(1, 64, 64, 256, 16, 256) > 0
    assert L.einx_lg_ws_bytes_heads(1, 64, 64, 256, 4, 256) == L.einx_lg_ws_bytes(1, 64, 64, 256, 256) > 0
    assert L.einx_lg_ws_bytes_heads(1, 64, 64, 256, 8, 256) < L.einx_lg_ws_bytes(1, 64, 64, 256, 256)  # narrower rotary table
