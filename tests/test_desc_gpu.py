"""GPU tests (-m gpu), component: desc.
SURVEY 8a rows A14-A16 and 8f-4 (sparse descriptor sampling, normalisation, upsampled dense maps, dense outputs on demand): desc.hip.
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

from helpers import (synth)
from gpu_support import (DESC, DEV, _four_pairs, _need_free_gb, _np, _sp_mnn_model, _t, _tiled, pkg)

pytestmark = pytest.mark.gpu


def test_desc_helpers_vs_golden_and_oracle(oracle):
    from importlib import import_module
    from test_oracle_golden import _desc_positions
    dd = import_module(pkg.__name__ + ".core.modules.utils.descriptor_util")
    for name in ("low_d32", "low_d256"):
        c = DESC.cases[name]
        raw = synth.normalish(c["seed"], (2, c["D"], c["hc"], c["wc"]))
        Hp, Wp = c["hc"] * 8, c["wc"] * 8
        idx = _desc_positions(c, 0)
        pos0 = np.stack([idx // Wp + 0.5, idx % Wp + 0.5, np.zeros(len(idx))], 1).astype(np.float32)
        out = dd.sparsify_low_resolution_descriptors(_t(raw), [_t(pos0), _t(pos0[:0])], (Hp, Wp), scale_factor=1.0)
        np.testing.assert_allclose(_np(out[0]), DESC[f"{name}.desc0"], atol=2e-6, rtol=0)
        assert tuple(out[1].shape) == tuple(DESC[f"{name}.desc1_shape"])
        exp = oracle.desc_sample_bilinear(raw, [idx, idx[:0]], (Hp, Wp), 1.0)
        assert np.array_equal(_np(out[0]), exp[0])
        co = dd.normalize_descriptors(_t(raw), 1.0)
        assert np.array_equal(_np(co), oracle.normalize_map(raw, 1.0))
        np.testing.assert_allclose(_np(co), DESC[f"{name}.coarse"], atol=2e-6, rtol=0)
    c = DESC.cases["full_d128"]
    raw = synth.normalish(c["seed"], (1, c["D"], c["H"], c["W"]))
    idx = _desc_positions(c, 0)
    pos0 = np.stack([idx // c["W"] + 0.5, idx % c["W"] + 0.5, np.zeros(len(idx))], 1).astype(np.float32)
    out = dd.sparsify_full_resolution_descriptors(_t(raw), (_t(pos0),), scale_factor=torch.tensor(1.41))
    np.testing.assert_allclose(_np(out[0]), DESC["full_d128.desc0"], atol=2e-6, rtol=0)
    assert np.array_equal(_np(out[0]), oracle.desc_gather(raw, [idx], 1.41)[0])
    c = DESC.cases["dense_d16"]
    raw = synth.normalish(c["seed"], (1, c["D"], c["hc"], c["wc"]))
    up = dd.upsample_descriptors(_t(raw), (c["hc"] * 8, c["wc"] * 8), 1.0)
    np.testing.assert_allclose(_np(up), DESC["dense_d16.up"], atol=2e-6, rtol=0)
    assert np.array_equal(_np(up), oracle.upsample_normalize(raw, (c["hc"] * 8, c["wc"] * 8), 1.0))


@pytest.mark.parametrize("shape", [(3, 256, 33, 44), (2, 128, 9, 70), (1, 320, 5, 7), (2, 7, 3, 41), (1, 600, 4, 9)],
                         ids=lambda s: "x".join(map(str, s)))
def test_normalize_map_tiles_and_channels_last_sampler(oracle, shape):
    """the LDS-tiled normalise (64 / 32 pixel tiles, ragged pixel and channel counts, D > 512 fall-back),
    its channels-last raw copy, and the sampler reading that copy: all bit-equal to the oracle."""
    N = pkg.native
    B, D, hc, wc = shape
    raw = synth.normalish(700 + D, shape)
    raw[0, :, 0, 0] = 0  # an all-zero pixel -> eps clamp
    t = _t(raw)
    if D <= 512:
        co, cl = N.normalize_map(t, 1.3, want_cl=True)
        assert np.array_equal(_np(cl), raw.reshape(B, D, hc * wc).transpose(0, 2, 1))
    else:
        co, cl = N.normalize_map(t, 1.3), None
    assert np.array_equal(_np(co), oracle.normalize_map(raw, 1.3))
    if D > 512:
        return
    Hp, Wp = hc * 8, wc * 8
    n = 50
    idx = [np.sort(np.argsort(synth.uniform01(800 + b, (Hp * Wp,)))[:n]).astype(np.int32) for b in range(B)]
    idx[-1] = idx[-1][:17]  # ragged count
    cap = n
    ind = np.zeros((B, cap), np.int32)
    for b in range(B):
        ind[b, :len(idx[b])] = idx[b]
    cnt = torch.tensor([len(i) for i in idx], dtype=torch.int32, device=DEV)
    exp = oracle.desc_sample_bilinear(raw, idx, (Hp, Wp), 1.0)
    for use_cl in (False, True):
        got = _np(N.desc_sample(t, _t(ind), cnt, (Hp, Wp), True, 1.0, raw_cl=cl if use_cl else None))
        for b in range(B):
            assert np.array_equal(got[b, :len(idx[b])], exp[b]), (use_cl, b)


@pytest.mark.parametrize("case", [
    (2, 16, 5, 7, 40, 56, (2, 3, 35, 50)),      # scale 1/8 with a crop window
    (1, 256, 33, 44, 264, 352, (2, 3, 260, 346)),  # the shipped geometry
    (2, 7, 5, 7, 33, 47, (0, 0, 33, 47)),       # non-integer scale, no crop
    (1, 9, 5, 7, 10, 14, (1, 2, 8, 11)),        # x2
    (1, 5, 6, 70, 120, 140, (3, 1, 110, 139)),  # 1/20 vertically: bands taller than one sweep; several column blocks
    (1, 4, 8, 9, 8, 9, (0, 0, 8, 9)),           # identity size
    (1, 6, 4, 60, 16, 360, (0, 0, 16, 360)),    # six column sweeps + 61-word coarse rows: 64.8 KB of dynamic LDS, still the two-kernel path
    (1, 6, 4, 63, 16, 378, (0, 0, 16, 378)),    # one word more than a launch may ask for: falls back to the band kernel
    (2, 33, 3, 5, 70, 40, (1, 0, 68, 40)),      # bands of 23 rows: three sweeps per band (extra units), 33 channels = a ragged channel group
], ids=lambda c: "x".join(map(str, c[:6])))
def test_upsample_normalize_bands(oracle, case):
    """upsample_descriptors + normalize + crop (dense outputs): band-wise kernel == per-pixel oracle, bit for bit."""
    B, D, hc, wc, Hp, Wp, (h0, w0, H, W) = case
    raw = synth.normalish(900 + D + hc, (B, D, hc, wc))
    got = _np(pkg.native.upsample_normalize(_t(raw), (Hp, Wp), (w0, Wp - w0 - W, h0, Hp - h0 - H), 1.25))
    exp = oracle.upsample_normalize(raw, (Hp, Wp), 1.25)[:, :, h0:h0 + H, w0:w0 + W]
    assert got.shape == exp.shape
    assert np.array_equal(got, exp)


def test_upsample_normalize_division_edge_cases(oracle):
    """The store kernel divides with two Newton corrections of v * (1/den) (exact for 2^-80 <= |v| <= den < 2^20) and falls
    back to IEEE divisions otherwise, per wave and channel: exact zeros (+-0), denormal-range values, huge norms and all-zero
    pixels (den clamps to 1e-12) must come out bit-equal to the oracle's plain `v / den` too."""
    B, D, hc, wc = 2, 40, 5, 7
    Hp, Wp = 40, 56
    raw = synth.normalish(977, (B, D, hc, wc)).astype(np.float32)
    raw[0, 3] = 0.0                 # a channel of exact zeros: v == 0 -> slow path for that channel
    raw[0, 5] = -0.0
    raw[0, 7] *= np.float32(1e-30)  # |v| < 2^-80
    raw[0, 9] *= np.float32(1e-42)  # denormal inputs
    raw[1, :, :2, :] = 0.0          # all-zero pixels: den = 1e-12, 0 / 1e-12
    raw[1, 11, 3:, :] *= np.float32(1e24)  # norms beyond 2^20: the whole sweep divides the IEEE way
    got = _np(pkg.native.upsample_normalize(_t(raw), (Hp, Wp), (3, 3, 2, 2), 1.41))
    exp = oracle.upsample_normalize(raw, (Hp, Wp), 1.41)[:, :, 2:Hp - 2, 3:Wp - 3]
    assert np.array_equal(got, exp)
    assert np.array_equal(np.signbit(got), np.signbit(exp))  # -0 stays -0


def test_batch_96_dense_descriptor_map_beyond_2_31_elements_equals_the_small_batch():
    _need_free_gb(48)
    B = 96
    assert B * 256 * 260 * 346 > 2**31
    model = _sp_mnn_model(dense_event=True)
    ev, mask, img = _four_pairs(4343)
    ef4, _, _ = model(_t(ev), _t(img), _t(mask))
    small = [ef4["normalized_descriptors"][r].clone() for r in range(4)]
    small_dd = [ef4["dense_descriptors"][r].clone() for r in range(4)]
    del ef4
    ef, _, _ = model(_t(_tiled(ev, B)), _t(_tiled(img, B)), _t(_tiled(mask, B)))
    nd = ef["normalized_descriptors"]
    assert tuple(nd.shape) == (B, 256, 260, 346)
    for b in (0, 1, 46, 47, 92, 93, 94, 95):  # 93 is the first image past 2^31 floats
        assert torch.equal(nd[b], small[b % 4]), f"image {b}: dense descriptor map"
        assert torch.equal(ef["dense_descriptors"][b], small_dd[b % 4]), f"image {b}: dense descriptor list entry"
    n = torch.linalg.vector_norm(nd[95], dim=0)
    scale = float(model.event_extractor.extractor.descriptor_scale_factor)
    assert float((n - scale).abs().max()) < 1e-4


def test_dense_outputs_on_demand_equal_the_eager_ones():
    """dense_outputs="lazy" (the package default): the dict carries the reference's dense keys from the start; their values
    are computed on first access -- through d[k], get, items, values, dict(d), {**d} alike -- and equal what the eager mode
    (dense_outputs=True) computes inside the forward."""
    from helpers import synth
    cfg = pkg.default_config("SP_MNN", event_channels=5)
    models = []
    for mode in ("lazy", True):
        m = pkg.EIM(cfg, device=DEV).eval()
        sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in m.state_dict().items()], seed=21)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        assert m.event_extractor.extractor.dense_outputs == "lazy"  # default
        for ext in (m.event_extractor.extractor, m.image_extractor.extractor):
            ext.dense_outputs = mode
        models.append(m)
    ev, mask = synth.synth_events(5, 2, 5, 96, 128)
    img = synth.synth_image(5, 2, 96, 128)
    le, li, _ = models[0](_t(ev), _t(img), _t(mask))
    ee, ei, _ = models[1](_t(ev), _t(img), _t(mask))
    dense_keys = ["dense_descriptors", "dense_positions", "normalized_descriptors"]
    for lazy, eager in ((le, ee), (li, ei)):
        assert sorted(lazy.keys()) == sorted(eager.keys())
        assert sorted(lazy.lazy_keys()) == dense_keys and eager.lazy_keys() == []
        assert torch.equal(lazy["sparse_positions"][0], eager["sparse_positions"][0]) and sorted(lazy.lazy_keys()) == dense_keys
    assert torch.equal(le["normalized_descriptors"], ee["normalized_descriptors"])  # d[k]
    assert sorted(le.lazy_keys()) == ["dense_descriptors", "dense_positions"]
    assert all(torch.equal(a, b) for a, b in zip(le.get("dense_descriptors"), ee["dense_descriptors"]))  # get
    assert all(torch.equal(a, b) for a, b in zip(dict(le.items())["dense_positions"], ee["dense_positions"]))  # items
    assert le.lazy_keys() == []
    plain = dict(li)  # CPython's dict() / {**d} merge goes through keys() + __getitem__ for this subclass
    assert torch.equal(plain["normalized_descriptors"], ei["normalized_descriptors"]) and li.lazy_keys() == []
    assert all(torch.equal(a, b) for a, b in zip({**li}["dense_positions"], ei["dense_positions"]))


def test_lazy_dense_and_forward_graph_on_the_silk_family():
    """The cell-1 networks (VGG_NP events + SiLK image): dense entries on demand (normalised full-resolution map, cropped) equal
    the eager ones, and forward_graph (no events mask given to the image side, 128-d descriptors) equals forward."""
    from helpers import synth
    cfg = pkg.default_config("SiLK_MNN", event_channels=5)
    model = pkg.EIM(cfg, device=DEV).eval()
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=29)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    ev, mask = synth.synth_events(70, 1, 5, 96, 128)
    img = synth.synth_image(70, 1, 96, 128)
    lazy = model(_t(ev), _t(img), _t(mask))
    assert sorted(lazy[1].lazy_keys()) == ["dense_descriptors", "dense_positions", "normalized_descriptors"]
    for ext in (model.event_extractor.extractor, model.image_extractor.extractor):
        ext.dense_outputs = True
    eager = model(_t(ev), _t(img), _t(mask))
    for side in (0, 1):
        assert torch.equal(lazy[side]["normalized_descriptors"], eager[side]["normalized_descriptors"])
        assert torch.equal(lazy[side]["dense_positions"][0], eager[side]["dense_positions"][0])
        assert torch.equal(lazy[side]["dense_descriptors"][0], eager[side]["dense_descriptors"][0])
    for ext in (model.event_extractor.extractor, model.image_extractor.extractor):
        ext.dense_outputs = "lazy"
    for it in range(2):
        g = model.forward_graph(_t(ev), _t(img), _t(mask))
        assert torch.equal(g[0]["sparse_descriptors"][0], lazy[0]["sparse_descriptors"][0])
        assert torch.equal(g[1]["sparse_positions"][0], lazy[1]["sparse_positions"][0])
        assert torch.equal(g[2]["matches0"][0], lazy[2]["matches0"][0])
        assert torch.equal(g[1]["normalized_descriptors"], eager[1]["normalized_descriptors"])
