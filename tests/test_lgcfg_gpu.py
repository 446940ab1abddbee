"""GPU tests (-m gpu): LightGlue configurations other than descriptor_dim 256 = 4 heads x 64.

The reference derives head_dim = descriptor_dim // num_heads and takes n_layers / input_dim from the conf
(core/modules/matchers/lightglue.py:246-248, 280-283, 456-461).  The kernels exist for 32-, 64- and 128-wide heads: the
shipped widths keep their own instantiations, every other configuration runs the same kernels with the widths as arguments.
Fixtures: tests/golden/lgcfg.npz, generated from the reference (gen_golden.py::gen_lgcfg)."""
import json
from importlib import import_module

import numpy as np
import pytest
import torch

from helpers import Golden, close_and_record, la_bound, lg_inputs, load_pkg, record_flips, state_dict_for

pytestmark = pytest.mark.gpu
pkg = load_pkg()
DEV = "cuda:0"
FTOL = 1e-4  # north_star: fp32 descriptors / scores within 1e-4
LGCFG = Golden("lgcfg")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _np(t):
    return t.detach().cpu().numpy()


def _conf(c):
    return {k: c[k] for k in ("input_dim", "descriptor_dim", "num_heads", "n_layers")}


def _model(c, keys=None):
    lg = pkg.LightGlue(_conf(c)).to(DEV)
    shapes = keys if keys is not None else {k: list(v.shape) for k, v in lg.state_dict().items()}
    sd = state_dict_for(dict(c, state_keys=shapes))
    lg.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return lg.eval(), sd


@pytest.mark.parametrize("fold", [True, False])
@pytest.mark.parametrize("name", list(LGCFG.cases))
def test_lightglue_other_widths_vs_reference_and_oracle(oracle, name, fold):
    c = LGCFG.cases[name]
    keys = json.loads(bytes(LGCFG[f"{name}.state_keys"]).decode())
    lg, sd = _model(c, keys)  # strict load: the parameter tree has the reference's names and shapes
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


BATCH_CASES = [
    # (conf, B, cap0, cap1): caps of 1024 with B >= 3 take the persistent 128x128-tile linears (>= 256 tiles), the small ones the
    # 64x64-tile kernels; d = 192 has a partial 128-column tile per q | k | v block; unequal caps run the two sides unstacked
    (dict(input_dim=256, descriptor_dim=256, num_heads=8, n_layers=2), 3, 1024, 1024),
    (dict(input_dim=128, descriptor_dim=192, num_heads=3, n_layers=2), 4, 1024, 1024),
    (dict(input_dim=256, descriptor_dim=512, num_heads=4, n_layers=1), 3, 1024, 640),
    (dict(input_dim=128, descriptor_dim=128, num_heads=4, n_layers=2), 5, 130, 130),
    (dict(input_dim=64, descriptor_dim=64, num_heads=2, n_layers=2), 2, 70, 200),
]


@pytest.mark.parametrize("case", range(len(BATCH_CASES)))
def test_lightglue_other_widths_batched_vs_per_pair_oracle(oracle, case):
    """Ragged batches through the device-count path (both kernels of the linears, stacked and unstacked sides): every pair
    equals its own oracle run; padding rows hold garbage and stay unmatched."""
    conf, B, cap0, cap1 = BATCH_CASES[case]
    bt = import_module(pkg.__name__ + ".core.modules.matchers._batched")
    c = dict(conf, wseed=700 + case)
    lg, sd = _model(c)
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
    lg, sd = _model(dict(conf, wseed=731))
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
    for conf in ({"input_dim": 60, "descriptor_dim": 60, "num_heads": 2}, {"input_dim": 512, "descriptor_dim": 512, "num_heads": 2}):
        with pytest.raises(NotImplementedError):  # head widths that are not a multiple of 4 / wider than 128 (16-wide heads run since round 6)
            pkg.LightGlue(conf)
    with pytest.raises(AssertionError):
        pkg.LightGlue({"input_dim": 256, "descriptor_dim": 256, "num_heads": 3})  # the reference's own assert (:247)
    L = import_module(pkg.__name__ + "._native").lib()
    assert L.einx_lg_ws_bytes_heads(1, 64, 64, 60, 2, 60) == 0 and L.einx_lg_ws_bytes_heads(1, 64, 64, 512, 2, 512) == 0
    assert L.einx_lg_ws_bytes_heads(1, 64, 64, 256, 16, 256) > 0
    assert L.einx_lg_ws_bytes_heads(1, 64, 64, 256, 4, 256) == L.einx_lg_ws_bytes(1, 64, 64, 256, 256) > 0
    assert L.einx_lg_ws_bytes_heads(1, 64, 64, 256, 8, 256) < L.einx_lg_ws_bytes(1, 64, 64, 256, 256)  # narrower rotary table
