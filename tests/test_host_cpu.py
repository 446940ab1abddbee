"""CPU-only checks of the host side: the C ABI is complete, the module tree reproduces the
reference's state_dict names/shapes, and the product refuses to run without a GPU (no fallback)."""
import json
import os
import re

import numpy as np
import pytest
import torch

from helpers import Golden, ROOT, load_pkg

pkg = load_pkg()


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "einx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(einx_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 18
    lib = pkg.native.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"libeinx_hip.so does not export {name}"
    # and the Python binding knows each of them
    from importlib import import_module
    sig = import_module(pkg.__name__ + "._lib").SIGNATURES
    assert declared == set(sig), declared ^ set(sig)
    assert lib.einx_version().startswith(b"einx-hip")


def test_no_gpu_means_no_silent_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    model = pkg.EIM(pkg.default_config("SP_MNN"), device="cpu").eval()
    ev = torch.zeros(1, 5, 64, 64)
    img = torch.zeros(1, 1, 64, 64)
    with pytest.raises(RuntimeError, match="HIP device"):
        model(ev, img, torch.zeros(1, 1, 64, 64, dtype=torch.bool))


E2E = Golden("e2e")
LG = Golden("lg")


@pytest.mark.parametrize("name", list(E2E.cases))
def test_state_dict_names_match_reference(name):
    c = E2E.cases[name]
    cfg = pkg.configs.to_attr(c["cfg"])
    model = pkg.EIM(cfg, device="cpu")
    mine = {k: list(v.shape) for k, v in model.state_dict().items()}
    ref = dict(c["state_keys"])
    # the generator skipped descriptor_scale_factor entries when synthesising weights
    mine_cmp = {k: v for k, v in mine.items() if not k.endswith("descriptor_scale_factor")}
    assert mine_cmp == ref
    assert any(k.endswith("descriptor_scale_factor") for k in mine)


@pytest.mark.parametrize("name", ["d256", "d128"])
def test_lightglue_state_dict(name):
    c = LG.cases[name]
    keys = json.loads(bytes(LG[f"{name}.state_keys"]).decode())
    lg = pkg.LightGlue({"input_dim": c["input_dim"]})
    mine = {k: list(v.shape) for k, v in lg.state_dict().items()}
    assert mine == keys


def test_unknown_types_raise_like_reference():
    cfg = pkg.default_config("SP_MNN")
    cfg.event_extractor.type = "nope"
    with pytest.raises(ValueError):
        pkg.EIM(cfg, device="cpu")
    cfg = pkg.default_config("SP_MNN")
    cfg.matcher.type = "nope"
    with pytest.raises(NotImplementedError):
        pkg.EIM(cfg, device="cpu")
    cfg = pkg.default_config("SP_MNN")
    cfg.name = "other"
    with pytest.raises(NotImplementedError):
        pkg.build_model(cfg, "cpu", None)


def test_padder_and_ranks():
    from importlib import import_module
    nat = pkg.native
    assert nat.padder_pads(260, 346, 8) == (3, 3, 2, 2)
    assert nat.padder_pads(260, 346, 1) == (0, 0, 0, 0)
    assert nat.topk_ranks(264 * 352, 1024) == (91903, 91904)
    assert nat.topk_capacity(264 * 352, 1024, 1.0) == 1024
    util = import_module(pkg.__name__ + ".core.modules.utils.util")
    p = util.Padder((1, 1, 260, 346), 8)
    x = torch.arange(260 * 346, dtype=torch.float32).reshape(1, 1, 260, 346)
    xp = p.pad(x)[0]
    assert xp.shape[-2:] == (264, 352)
    assert torch.equal(p.unpad(xp)[0], x)


def test_parent_load_state_dict_invalidates_native_weight_images():
    """nn.Module.load_state_dict on a PARENT recurses through _load_from_state_dict and never
    calls the child's load_state_dict override; the cached kernel-native images must still be dropped."""
    model = pkg.EIM(pkg.default_config("SP_LG"), device="cpu").eval()
    ext = model.image_extractor.extractor
    lg = model.matcher.matcher
    sentinel = object()
    ext._engine, ext._scale_host, lg._packed = sentinel, 1.0, sentinel
    model.load_state_dict(model.state_dict())
    assert ext._engine is None and ext._scale_host is None and lg._packed is None
    # in-place edits made through the parameter (optimiser steps, `with no_grad(): p.add_()`) change the signature
    ev = model.event_extractor.extractor
    s0 = ev._signature()
    with torch.no_grad():
        next(ev.parameters()).add_(1.0)
    assert ev._signature() != s0


def test_data_alias_edits_need_refresh():
    """`p.data` is an alias with its own version counter, so `p.data.copy_()` / `p.data.mul_()` leave `p._version`
    (what the HOST-side weight-image caches key on) unchanged: refresh() (and a parent's load_state_dict) drops every native
    image.  Since round 4 the device-side content watch catches such edits at the next forward (tests/test_boundary_gpu.py); this
    test pins the torch semantics that make the watch necessary."""
    model = pkg.EIM(pkg.default_config("SP_LG"), device="cpu").eval()
    ev, lg = model.event_extractor.extractor, model.matcher.matcher
    p = next(ev.parameters())
    s0, v0 = ev._signature(), p._version
    p.data.mul_(2.0)
    p.data.copy_(torch.zeros_like(p))
    assert p._version == v0 and ev._signature() == s0  # torch semantics the contract rests on (if this ever changes, auto-detect)
    sentinel = object()
    ev._engine, ev._scale_host, lg._packed = sentinel, 1.0, sentinel
    ev.refresh()
    lg.refresh()
    assert ev._engine is None and ev._scale_host is None and lg._packed is None and lg._sig_tensors is None
    with torch.no_grad():
        p.add_(1.0)  # the supported in-place form IS detected
    assert ev._signature() != s0


def test_native_wrappers_reject_wrong_dtypes():
    """raw-pointer kernels assume fp32 / int32; anything else must raise, not be misread."""
    nat = pkg.native

    class FakeCuda:  # a tensor that claims to be on a HIP device, enough for the checks that run before any launch
        def __init__(self, t):
            self.t = t
            self.device = torch.device("cuda", 0)
            self.dtype = t.dtype

        def is_contiguous(self):
            return True

    for bad in (torch.float16, torch.bfloat16, torch.float64, torch.int64):
        with pytest.raises(TypeError, match="expected a torch.float32"):
            nat._dev_check(FakeCuda(torch.zeros(4, dtype=bad)))
    with pytest.raises(TypeError, match="expected a torch.int32"):
        nat._dev_check(FakeCuda(torch.zeros(4, dtype=torch.int64)), dt=torch.int32)
    nat._dev_check(FakeCuda(torch.zeros(4)), None)
    with pytest.raises(RuntimeError, match="HIP device"):
        nat._dev_check(torch.zeros(4))


def test_entry_points_switch_to_the_inputs_device(monkeypatch):
    """ctypes launches need the inputs' device to be current (torch's own operators switch per call, so the reference works
    with a model on cuda:1 while cuda:0 is current): the entry-point decorator enters torch.cuda.device(inputs' device) when
    it differs from the current one, and does nothing otherwise or for CPU inputs."""
    nat = pkg._native
    entered = []

    class FakeGuard:
        def __init__(self, dev):
            self.dev = dev

        def __enter__(self):
            entered.append(self.dev)

        def __exit__(self, *a):
            return False

    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(torch.cuda, "device", FakeGuard)
    seen = {}
    monkeypatch.setattr(nat, "_first_device", lambda obj, depth=0: seen.get(id(obj)))

    class M:
        @nat.on_input_device
        def forward(self, x, y=None):
            return "ran"

    a, b = object(), object()
    seen[id(a)] = torch.device("cuda", 1)
    seen[id(b)] = torch.device("cuda", 0)
    assert M().forward(a) == "ran" and entered == [torch.device("cuda", 1)]
    assert M().forward(b) == "ran" and len(entered) == 1  # already current: no guard
    assert M().forward(object(), y=a) == "ran" and len(entered) == 2  # found among the keyword arguments
    assert M().forward(torch.zeros(2)) == "ran" and len(entered) == 2  # CPU input: left to the wrappers' own error


def test_nms_pass_budget_grows_and_decays(monkeypatch):
    """ExtractorEngine's wide-pass budget (host logic, no GPU): EINX_NMS_PASSES <= 0 cannot disable the growth,
    a retry quadruples the budget, 64 converged forwards halve it again, never below the base."""
    import importlib
    ex = importlib.import_module(pkg.__name__ + "._extract")
    for env, base in (("0", 1), ("-3", 1), ("8", 8)):
        monkeypatch.setenv("EINX_NMS_PASSES", env)
        eng = ex.ExtractorEngine("vgg", top_k=1024, radius=4, border=4, det_thr=1.0, ordering="yx", cell=8)
        assert eng.nms_base == base and eng.nms_iters == base
        assert eng.grow_nms_iters() == 4 * base and eng.grow_nms_iters() == 16 * base
        for _ in range(63):
            eng.note_converged()
        assert eng.nms_iters == 16 * base
        eng.note_converged()
        assert eng.nms_iters == 8 * base
        for _ in range(64 * 8):
            eng.note_converged()
        assert eng.nms_iters == base


def test_feats_dict_resolves_lazy_entries_on_every_read_path():
    """_extract.FeatsDict (dense outputs on demand): a lazy entry is computed once, by whichever read reaches it first --
    d[k], get, items, values, pop, popitem, setdefault, copy, `|`, `==`, and CPython's own merges dict(d) / {**d} / other.update(d)."""
    import importlib
    ex = importlib.import_module(pkg.__name__ + "._extract")

    def fresh():
        d = ex.FeatsDict()
        calls = []
        dict.__setitem__(d, "score", 1)
        dict.__setitem__(d, "normalized_descriptors", ex._Lazy(lambda dd: calls.append(dd is d) or 42))
        return d, calls

    for read in (lambda d: d["normalized_descriptors"], lambda d: d.get("normalized_descriptors"), lambda d: dict(d.items())["normalized_descriptors"],
                 lambda d: list(d.values())[1], lambda d: d.pop("normalized_descriptors"), lambda d: d.setdefault("normalized_descriptors", 7),
                 lambda d: d.copy()["normalized_descriptors"], lambda d: dict(d)["normalized_descriptors"], lambda d: {**d}["normalized_descriptors"],
                 lambda d: (lambda o: (o.update(d), o)[1])({})["normalized_descriptors"],
                 # the remaining dict entry points no longer leak the sentinel
                 lambda d: d.popitem()[1], lambda d: (d | {"x": 0})["normalized_descriptors"], lambda d: ({"x": 0} | d)["normalized_descriptors"],
                 lambda d: 42 if d == {"score": 1, "normalized_descriptors": 42} else None,
                 lambda d: 42 if not (d != {"score": 1, "normalized_descriptors": 42}) else None):
        d, calls = fresh()
        assert sorted(d.keys()) == ["normalized_descriptors", "score"] and "normalized_descriptors" in d and len(d) == 2
        assert d.lazy_keys() == ["normalized_descriptors"] and calls == []
        assert read(d) == 42 and calls == [True]
        if "normalized_descriptors" in d:
            assert d["normalized_descriptors"] == 42 and calls == [True] and d.lazy_keys() == []  # computed once
    d, _ = fresh()
    assert d.get("missing", 5) == 5 and d.pop("missing", 6) == 6


def test_pooled_padding0_constructs_and_raises_what_the_reference_raises():
    """VGGExtractor(padding=0) (pooled): the reference constructs it, and every forward of it raises IndexError -- with a mask at
    `score[~score_mask] = 0`, without one in filter_sparse_feats (recorded from the reference: tests/golden/pad0_pooled.json).  The
    drop-in used to refuse the constructor argument (VERDICT r4 "missing" 2); it now constructs (same state_dict keys) and raises the
    same exception type and message, before any device work."""
    import json
    import torch
    from helpers import GOLDEN, load_pkg
    pkg = load_pkg()
    from importlib import import_module
    ee = import_module(pkg.__name__ + ".core.modules.event_extractors.EventExtractors")
    rec = json.load(open(os.path.join(GOLDEN, "pad0_pooled.json")))
    m = ee.VGGExtractor(in_channels=5, feat_channels=128, descriptor_dim=256, nms_radius=4, detection_top_k=50, detection_threshold=1.0, padding=0).eval()
    assert len(m.state_dict()) == rec["state_keys"]
    assert len(rec["cases"]) == 3
    for c in rec["cases"]:
        x = torch.zeros((c["B"], 5, c["H"], c["W"]))
        mask = torch.ones((c["B"], 1, c["H"], c["W"]), dtype=torch.bool)
        for tag, args in (("no_mask", (x,)), ("mask", (x, mask))):
            assert c[tag]["type"] == "IndexError"
            with pytest.raises(IndexError) as ei:
                m(*args)
            assert str(ei.value) == c[tag]["message"]


def test_lightglue_other_widths_have_the_references_parameter_tree():
    """descriptor_dim / num_heads / n_layers / input_dim from the conf (lightglue.py:446-466): names and shapes of every
    parameter equal the reference's (tests/golden/lgcfg.npz); add_scale_ori widens posenc.Wr to 4 inputs; head widths that are
    not a multiple of 4 or exceed 256 are refused at construction."""
    from helpers import Golden
    g = Golden("lgcfg")
    for name, c in g.cases.items():
        keys = json.loads(bytes(g[f"{name}.state_keys"]).decode())
        lg = pkg.LightGlue({k: c[k] for k in ("input_dim", "descriptor_dim", "num_heads", "n_layers")})
        assert {k: list(v.shape) for k, v in sorted(lg.state_dict().items())} == keys, name
    aso = pkg.LightGlue({"input_dim": 256, "add_scale_ori": True})
    assert {k: list(v.shape) for k, v in aso.state_dict().items() if k.startswith("posenc")} == g.meta["add_scale_ori"]["state_keys"]
    pkg.LightGlue({"descriptor_dim": 256, "num_heads": 16})  # 16-wide heads: accepted since round 6 (zero-padded to 32 in the attention)
    for conf in ({"descriptor_dim": 60, "num_heads": 2, "input_dim": 60}, {"descriptor_dim": 1024, "num_heads": 2, "input_dim": 1024}):
        with pytest.raises(NotImplementedError):  # head widths that are not a multiple of 4, or wider than 256
            pkg.LightGlue(conf)
    with pytest.raises(AssertionError):
        pkg.LightGlue({"descriptor_dim": 256, "num_heads": 3})


def test_events_pack_host_helper_converts_and_concatenates_like_numpy():
    """einx_events_pack (host side of libeinx_hip.so, no kernel): B ragged samples with the element types datasets hand out
    (float32 / float64 / int64 / uint16 / bool ...) -> the flat fp32 x / y / p and fp64 t arrays + offsets, equal to
    np.concatenate(...).astype(...) for every thread count."""
    import ctypes
    from importlib import import_module
    _lib = import_module(pkg.__name__ + "._lib")
    L = pkg.native.lib()
    rng = np.random.default_rng(3)
    codes = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.int64): 2, np.dtype(np.int32): 3, np.dtype(np.int16): 4,
             np.dtype(np.uint16): 5, np.dtype(np.int8): 6, np.dtype(np.uint8): 7, np.dtype(np.uint32): 8, np.dtype(np.uint64): 9, np.dtype(np.bool_): 7}
    xt = [np.float32, np.uint16, np.int64, np.float64, np.int32, np.int16, np.uint32, np.uint64]
    pt = [np.float32, np.bool_, np.int8, np.uint8, np.float64]
    evs = []
    for b, n in enumerate([0, 1, 70001, 5, 65536, 131073, 0, 999]):
        t = 1.5e9 + np.cumsum(rng.random(n))
        evs.append({"x": rng.integers(0, 346, n).astype(xt[b % len(xt)]), "y": rng.integers(0, 260, n).astype(xt[(b + 3) % len(xt)]),
                    "t": t if b % 2 else t.astype(np.float32), "p": (rng.random(n) < 0.5).astype(pt[b % len(pt)])})
    B = len(evs)
    arr = (_lib.EventArrays * B)()
    for b, e in enumerate(evs):
        arr[b] = _lib.EventArrays(e["x"].ctypes.data, e["y"].ctypes.data, e["t"].ctypes.data, e["p"].ctypes.data, codes[e["x"].dtype],
                                  codes[e["y"].dtype], codes[e["t"].dtype], codes[e["p"].dtype], len(e["x"]))
    N = sum(len(e["x"]) for e in evs)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    for threads in (1, 3, 8):
        x, y, p = (np.full(N, -7, np.float32) for _ in range(3))
        t = np.full(N, -7, np.float64)
        offs = np.full(B + 1, -1, np.int64)
        assert L.einx_events_pack(arr, B, P(x), P(y), P(t), P(p), P(offs), threads) == 0
        assert offs.tolist() == np.concatenate([[0], np.cumsum([len(e["x"]) for e in evs])]).tolist()
        for name, got, dt in (("x", x, np.float32), ("y", y, np.float32), ("t", t, np.float64), ("p", p, np.float32)):
            assert np.array_equal(got, np.concatenate([e[name].astype(dt) for e in evs])), (name, threads)
    bad = _lib.EventArrays(0, 0, 0, 0, 0, 0, 99, 0, 0)
    assert L.einx_events_pack(ctypes.byref(bad), 1, None, None, None, None, P(offs), 1) != 0  # unknown element type
