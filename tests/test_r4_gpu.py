"""Round-4 GPU parity tests (-m gpu): slow-converging NMS maps (device finisher bound, host retry), build-flag guard."""
import numpy as np
import pytest
import torch

from helpers import load_pkg

pytestmark = pytest.mark.gpu
pkg = load_pkg()
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _np(t):
    return t.detach().cpu().numpy()


def _ramp(H, W):
    s = (np.arange(W, dtype=np.float32)[None, :] + 1) / np.float32(W + 1)
    return np.broadcast_to(s, (H, W)).copy()[None, None]


def _serpentine(H, W):
    """values increasing along a boustrophedon path through EVERY pixel: each maximum is decided only after the one that
    follows it on the path, 380 passes on 96x96"""
    m = np.zeros((H, W), np.float32)
    v = 1
    for y in range(2, H - 2):
        for x in (range(W) if y % 2 == 0 else range(W - 1, -1, -1)):
            m[y, x] = np.float32(v) / np.float32((H - 4) * W + 1)
            v += 1
    return m[None, None]


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


def test_shipped_library_is_not_a_timing_only_build():
    assert pkg.native.lib().einx_build_flags() == b""


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


@pytest.mark.parametrize("cfg_name,B", [("SP_MNN", 1), ("SP_LG", 1), ("SP_MNN", 3)])
def test_forward_graph_equals_forward(cfg_name, B):
    """EIM.forward_graph (one hipGraph launch per forward) returns what EIM.forward returns, call after call, also when the
    inputs change between calls and when the dense entries are read on demand."""
    from helpers import synth
    cfg = pkg.default_config(cfg_name, event_channels=5)
    model = pkg.EIM(cfg, device=DEV).eval()
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=23)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    for it in range(3):
        ev, mask = synth.synth_events(40 + it, B, 5)
        img = synth.synth_image(40 + it, B)
        ef, imf, m = model(_t(ev), _t(img), _t(mask))
        exp = {"pos": [p.clone() for p in ef["sparse_positions"]], "desc": [d.clone() for d in imf["sparse_descriptors"]],
               "m0": [t.clone() for t in m["matches0"]], "mk": [t.clone() for t in m["matched_kpts1"]],
               "nd": ef["normalized_descriptors"].clone() if it == 1 else None, "score": imf["score"].clone()}
        img_t = _t(img)
        gf, gi, gm = model.forward_graph(_t(ev), img_t, _t(mask))
        assert np.array_equal(_np(img_t), img)  # the graph scales its own copy of the image
        for b in range(B):
            assert torch.equal(gf["sparse_positions"][b], exp["pos"][b]) and torch.equal(gi["sparse_descriptors"][b], exp["desc"][b])
            assert torch.equal(gm["matches0"][b], exp["m0"][b]) and torch.equal(gm["matched_kpts1"][b], exp["mk"][b])
        assert torch.equal(gi["score"], exp["score"])
        assert sorted(gf.keys()) == sorted(ef.keys())
        if exp["nd"] is not None:
            assert torch.equal(gf["normalized_descriptors"], exp["nd"])  # lazy entry resolved against THIS replay's buffers
    assert len(model._graphs) == 1


def test_forward_graph_budget_fallback_stays_on_the_graphs_buffers():
    """when a replay exceeds the captured NMS pass budget the forward is finished eagerly on the graph's OWN buffers
    (the caller's image is never scaled in place), equals the eager forward, and the graph is dropped so that the next call
    captures one with the grown budget.  nms_radius 3: the generic pass kernel (radius 4 finishes on the device); the budget
    is lowered to ONE pass and the graph captured on fully masked inputs (empty score maps converge at once), so an ordinary
    pair (3 passes) exceeds it."""
    from helpers import synth
    cfg = pkg.default_config("SP_MNN", event_channels=5)
    cfg.event_extractor.vgg.nms_radius = 3
    cfg.image_extractor.superpointv1.nms_radius = 3

    def build():
        m = pkg.EIM(cfg, device=DEV).eval()
        sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in m.state_dict().items()], seed=29)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        for w in (m.event_extractor, m.image_extractor):
            eng = w.extractor.engine()
            eng.nms_base = eng.nms_iters = 1
        return m

    model = build()
    ev, mask = synth.synth_events(70, 1, 5)
    img = synth.synth_image(70, 1)
    none = np.zeros_like(mask)
    allm = np.ones_like(mask)
    model.forward_graph(_t(ev), _t(img), _t(none), _t(none))  # capture: nothing to suppress, one pass is enough
    assert len(model._graphs) == 1
    assert model.image_extractor.extractor.engine().nms_iters == 1
    exp = build()(_t(ev), _t(img.copy()), _t(mask), _t(allm))
    img_t = _t(img)
    got = model.forward_graph(_t(ev), img_t, _t(mask), _t(allm))  # an ordinary pair: more than one pass
    assert model.image_extractor.extractor.engine().nms_iters > 1, "the pair did not exceed the captured budget"
    assert np.array_equal(_np(img_t), img)  # the caller's tensor is untouched (round 4 divided it by 255 on this path)
    assert len(model._graphs) == 0  # dropped: it holds the old budget
    for side in (0, 1):
        assert got[side]["sparse_positions"][0].shape[0] > 100
        assert torch.equal(got[side]["sparse_positions"][0], exp[side]["sparse_positions"][0])
        assert torch.equal(got[side]["sparse_descriptors"][0], exp[side]["sparse_descriptors"][0])
        assert torch.equal(got[side]["nms"], exp[side]["nms"])
    assert torch.equal(got[2]["matches0"][0], exp[2]["matches0"][0])
    again = model.forward_graph(_t(ev), _t(img), _t(mask), _t(allm))  # new capture with the grown budget: no fallback now
    assert len(model._graphs) == 1
    assert torch.equal(again[1]["sparse_positions"][0], exp[1]["sparse_positions"][0])
    # a configuration change is part of the cache key: no stale replay
    model.matcher.matcher.want_log_assignment = False
    model.forward_graph(_t(ev), _t(img), _t(mask), _t(allm))
    assert len(model._graphs) == 2


@pytest.mark.parametrize("cfg_name", ["SP_MNN", "SP_LG"])
def test_data_edits_of_weights_take_effect_at_the_next_forward(cfg_name):
    """The reference's modules are plain nn.Modules: `p.data.mul_(..)` / `p.data.copy_(..)` change the next forward.  Here weights
    are repacked / folded into native images and no host-side version counter sees a `.data` edit; the device-side content watch
    (einx_params_hash: every word hashed since round 5, read back with the counts) does, and the forward that notices rebuilds the
    images and runs again -- the result equals a model built from the edited weights."""
    from helpers import synth
    cfg = pkg.default_config(cfg_name, event_channels=5)

    def build(sd):
        m = pkg.EIM(cfg, device=DEV).eval()
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        for ext in (m.event_extractor.extractor, m.image_extractor.extractor):
            ext.dense_outputs = False
        return m

    model0 = pkg.EIM(cfg, device=DEV)
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model0.state_dict().items()], seed=31)
    model = build(sd)
    ev, mask = synth.synth_events(60, 2, 5)
    img = synth.synth_image(60, 2)
    before = model(_t(ev), _t(img), _t(mask))
    edits = ["event_extractor.extractor.backbone.l1.1.0.weight", "image_extractor.extractor.convDb.bias"]
    if cfg_name == "SP_LG":
        edits.append("matcher.matcher.transformers.3.self_attn.ffn.3.weight")
    params = dict(model.named_parameters())
    sd2 = dict(sd)
    for k in edits:
        new = (sd[k] * np.float32(1.25) + np.float32(0.01)).astype(np.float32)
        params[k].data.copy_(torch.from_numpy(new).to(DEV))  # the edit no version counter sees
        sd2[k] = new
    img_t = _t(img)
    got = model(_t(ev), img_t, _t(mask))
    exp = build(sd2)(_t(ev), _t(img), _t(mask))
    assert np.array_equal(_np(img_t), img / np.float32(255.0))  # scaled in place exactly once although the forward ran twice
    changed = False
    for b in range(2):
        for side in (0, 1):
            assert torch.equal(got[side]["sparse_positions"][b], exp[side]["sparse_positions"][b])
            assert torch.equal(got[side]["sparse_descriptors"][b], exp[side]["sparse_descriptors"][b])
            changed |= not torch.equal(got[side]["sparse_descriptors"][b], before[side]["sparse_descriptors"][b]) \
                if got[side]["sparse_descriptors"][b].shape == before[side]["sparse_descriptors"][b].shape else True
        assert torch.equal(got[2]["matches0"][b], exp[2]["matches0"][b])
        assert torch.equal(got[2]["matching_scores0"][b], exp[2]["matching_scores0"][b])
    assert changed
    again = model(_t(ev), _t(img), _t(mask))  # steady state: nothing stale any more, same result
    assert torch.equal(again[2]["matches0"][0], exp[2]["matches0"][0])
    # the standalone extractor front-end notices as well
    ext = model.image_extractor.extractor
    f0 = ext(_t(img) / 255.0 if False else _t(img))
    ext.convDb.bias.data.add_(0.5)
    img2 = _t(img)
    f1 = ext(img2)
    assert np.array_equal(_np(img2), img / np.float32(255.0))
    assert not torch.equal(f1["raw_descriptors"], f0["raw_descriptors"])
    assert torch.allclose(f1["raw_descriptors"], f0["raw_descriptors"] + 0.5, atol=1e-6)


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
