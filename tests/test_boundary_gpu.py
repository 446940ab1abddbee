"""GPU tests (-m gpu), component: boundary.
SURVEY 8b: the drop-in boundary -- Python API quirks, the C ABI and its torch-free host, graphs / streams, weight edits behind the native images, library-owned state, input layouts.
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

from helpers import (ROOT, rgb_input, state_dict_for, sub_dict, synth)
from gpu_support import (DEV, MNN, RGB, _eim_model, _feats_equal_oracle, _np, _t, _with_layout, pkg)

pytestmark = pytest.mark.gpu


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


def test_shipped_library_is_not_a_timing_only_build():
    assert pkg.native.lib().einx_build_flags() == b""


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


def test_no_forward_stalls_after_a_weight_reload():
    """The reference's evaluation scripts load a checkpoint and then call the model pair by pair
    (test_events-image_same-time.py:109-194).  Rounds 3-4 saw one-off 30-80 ms stalls of single forwards in the first ~15
    forwards after a LightGlue weight reload ("host stalled inside hipLaunchKernel, device idle").  Round 5 found the cause
    outside the library: CFS bandwidth throttling of the whole container -- CPU thread pools sized from the 256 visible CPUs
    (OpenMP 128, OpenBLAS 64) under a 16-CPU cgroup quota spin-wait after host-side parallel regions, exhaust the quota, and the
    kernel freezes every thread until the next 100 ms period (profiles/r05_notes.md).  With the pools sized to the quota
    (tests/conftest.py, bench.py::main, placement.cap_thread_pools) a reload is followed by ordinary forwards: host-side
    linear algebra + reload, then no forward of the next 20 takes more than 3x the median."""
    from helpers import synth
    from conftest import HOST_THREADS
    assert torch.get_num_threads() <= max(HOST_THREADS, 1)
    cfg = pkg.default_config("SP_LG", event_channels=5)
    model = pkg.EIM(cfg, device=DEV).eval()
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=37)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    ev, mask = synth.synth_events(90, 1, 5)
    img = synth.synth_image(90, 1)
    evt, mt, src = _t(ev), _t(mask), _t(img)
    buf = torch.empty_like(src)

    def step():
        buf.copy_(src)
        model(evt, buf, mt)
        torch.cuda.synchronize()

    for _ in range(5):
        step()
    worst = []
    for rep in range(3):
        # what bench.py's calibration does in front of its reload: host-side parallel regions (torch CPU operators, BLAS) ...
        a = np.random.default_rng(rep).standard_normal((2048, 256)).astype(np.float32)
        np.linalg.svd(a, full_matrices=False)
        t_ = torch.from_numpy(a)
        (t_ @ t_.T).sum().item()
        # ... then the reload itself: the matcher's weights change, its native images are rebuilt at the next forward
        new = {k: torch.from_numpy(v * np.float32(1.0 + 0.01 * (rep + 1))) for k, v in sd.items() if k.startswith("matcher.") and v.dtype == np.float32}
        model.load_state_dict(new, strict=False)
        step()  # rebuilds the images (not timed: it does real work)
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            step()
            ts.append((time.perf_counter() - t0) * 1e3)
        med = statistics.median(ts)
        worst.append((max(ts), med))
        assert max(ts) <= 3.0 * med, f"a forward after the reload took {max(ts):.1f} ms (median {med:.2f} ms): {[round(t, 1) for t in ts]}"
    print("post-reload forwards (max, median) ms:", [(round(a_, 2), round(b_, 2)) for a_, b_ in worst])


@pytest.mark.parametrize("with_lg", [False, True])
def test_torch_free_c_abi_host_values_equal_the_oracle(oracle, tmp_path, with_lg):
    """examples/c_abi_host -- the only consumer of include/einx.h that is neither Python nor torch -- fed with a seeded state
    dict and seeded inputs through a file: its keypoints, descriptors and match indices are bit-equal to the oracle's
    (round 4 only checked its exit status and three log lines).  with_lg: the program also fills einx_lg_weights / einx_lg_layer
    from C and runs a 3-layer LightGlue on the same features: assignments equal the oracle's, scores within 1e-4."""
    from helpers import sub_dict, synth
    exe = os.path.join(ROOT, "examples", "c_abi_host")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    B, H, W, CE = 2, 260, 346, 5
    cfg = pkg.default_config("SP_MNN", event_channels=CE)
    model = pkg.EIM(cfg, device=DEV)
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=41)
    ev, mask = synth.synth_events(95, B, CE)
    img = synth.synth_image(95, B)
    esd, isd = sub_dict(sd, "event_extractor.extractor."), sub_dict(sd, "image_extractor.extractor.")
    blob = []

    def vgg_block(conv, bn):  # the program's drawing order: w, b, gamma, beta, mean, var
        blob.extend([esd[conv + ".weight"], esd[conv + ".bias"]])
        blob.extend([esd[f"{bn}.{leaf}"] for leaf in ("weight", "bias", "running_mean", "running_var")])

    for s_ in range(1, 5):
        for j in (0, 1):
            vgg_block(f"backbone.l{s_}.{j}.0", f"backbone.l{s_}.{j}.2")
    vgg_block("detector_head._detH1.0", "detector_head._detH1.2")
    vgg_block("detector_head._detH2.0", "detector_head._detH2.1")
    vgg_block("descriptor_head._desH1.0", "descriptor_head._desH1.2")
    vgg_block("descriptor_head._desH2.0", "descriptor_head._desH2.1")
    for name in ("conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convPa", "convPb", "convDa", "convDb"):
        blob.extend([isd[name + ".weight"], isd[name + ".bias"]])
    blob.extend([ev, mask.astype(np.uint8), img])
    lgsd = None
    if with_lg:  # the fields of einx_lg_layer in their order (matrix, bias), then posenc.Wr, final_proj, matchability
        lg = pkg.LightGlue({"input_dim": 256, "n_layers": 3})
        lgsd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in lg.state_dict().items()], seed=43)
        for i in range(3):
            for blk, names in ((f"transformers.{i}.self_attn.", ("Wqkv", "out_proj", "ffn.0", "ffn.1", "ffn.3")),
                               (f"transformers.{i}.cross_attn.", ("to_qk", "to_v", "to_out", "ffn.0", "ffn.1", "ffn.3"))):
                for nm in names:
                    blob.extend([lgsd[blk + nm + ".weight"], lgsd[blk + nm + ".bias"]])
        blob.append(lgsd["posenc.Wr.weight"])
        for nm in ("final_proj", "matchability"):
            blob.extend([lgsd[f"log_assignment.2.{nm}.weight"], lgsd[f"log_assignment.2.{nm}.bias"]])
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        for a in blob:
            f.write(np.ascontiguousarray(a).tobytes())
    r = subprocess.run([exe, str(B), fin, fout] + (["lg"] if with_lg else []), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "C ABI host: OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    raw = open(fout, "rb").read()
    cap, D, off = 1024, 256, 0

    def take(dtype, shape):
        nonlocal off
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        a = np.frombuffer(raw[off:off + n], dtype).reshape(shape)
        off += n
        return a

    sides = []
    for _ in range(2):
        sides.append((take(np.int32, (B,)), take(np.float32, (B, cap, 3)), take(np.float32, (B, cap, D))))
    m0, nmatch = take(np.int64, (B, cap)), take(np.int32, (B,))
    if with_lg:
        lm0, lm1, ls0 = take(np.int64, (B, cap)), take(np.int64, (B, cap)), take(np.float32, (B, cap))
    assert off == len(raw)
    ecfg, icfg = cfg.event_extractor.vgg, cfg.image_extractor.superpointv1
    oe = oracle.extractor_forward("vgg", esd, ev.copy(), mask, top_k=1024, radius=4, border=4, det_thr=1.0, scale=1.0)
    oi = oracle.extractor_forward("superpointv1", isd, img.copy(), None, top_k=1024, radius=4, border=4, det_thr=1.0, scale=1.0)
    assert ecfg.nms_radius == 4 and icfg.remove_borders == 4  # the example hard-codes the shipped settings
    for (cnt, pos, desc), exp in zip(sides, (oe, oi)):
        for b in range(B):
            n = len(exp["sparse_positions"][b])
            assert cnt[b] == n > 900
            assert np.array_equal(pos[b, :n], exp["sparse_positions"][b])
            assert np.array_equal(desc[b, :n], exp["sparse_descriptors"][b])
    for b in range(B):
        em = oracle.mnn(oe["sparse_descriptors"][b], oi["sparse_descriptors"][b], want_la=False)
        n = len(oe["sparse_positions"][b])
        assert np.array_equal(m0[b, :n], em["matches0"])
        assert nmatch[b] == int((em["matches0"] > -1).sum())
        if with_lg:
            m = len(oi["sparse_positions"][b])
            el = oracle.lightglue(lgsd, oe["sparse_positions"][b], oe["sparse_descriptors"][b], oi["sparse_positions"][b],
                                  oi["sparse_descriptors"][b], n_layers=3, heads=4)
            assert np.array_equal(lm0[b, :n], el["matches0"]) and np.array_equal(lm1[b, :m], el["matches1"])
            assert np.abs(ls0[b, :n] - el["matching_scores0"]).max() <= 1e-4
            assert f"pair {b}: {int((el['matches0'] > -1).sum())} LightGlue matches" in r.stdout


@pytest.mark.parametrize("cfg_name", ["SP_MNN", "SP_LG"])
def test_single_element_data_edits_are_seen_by_the_weight_watch(cfg_name):
    """Round 4's content watch hashed 65 sampled words per tensor: `p.data[i, j, ...] = v` at an unsampled position was silently
    ignored.  The watch now hashes every word (one wave per 4096-word row): ONE edited element anywhere in a
    convolution weight, a BatchNorm buffer or a LightGlue matrix -- at positions the old sampler provably skipped -- makes the next
    forward rebuild the native images, and the result equals a model built from the edited weights."""
    from helpers import synth
    cfg = pkg.default_config(cfg_name, event_channels=5)

    def build(sd):
        m = pkg.EIM(cfg, device=DEV).eval()
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        return m

    model0 = pkg.EIM(cfg, device=DEV)
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model0.state_dict().items()], seed=43)
    model = build(sd)
    ev, mask = synth.synth_events(61, 1, 5)
    img = synth.synth_image(61, 1)
    model(_t(ev), _t(img), _t(mask))
    edits = ["event_extractor.extractor.backbone.l3.0.0.weight", "event_extractor.extractor.backbone.l2.1.2.running_var",
             "image_extractor.extractor.conv4a.weight"]
    if cfg_name == "SP_LG":
        edits.append("matcher.matcher.transformers.5.cross_attn.to_out.weight")
    tensors = dict(model.named_parameters())
    tensors.update(dict(model.named_buffers()))
    sd2 = dict(sd)
    for k in edits:
        n = sd[k].size
        # round 4 sampled words floor(lane * n / 64) for lane = 0..63 and the last word: pick a flat index that is none of them
        sampled = {(lane * n) // 64 for lane in range(64)} | {n - 1}
        flat = next((i for i in range(n // 3, n) if i not in sampled), n // 2)  # (tensors of <= 64 words were covered entirely)
        new = sd[k].copy().reshape(-1)
        new[flat] = new[flat] * np.float32(-3.0) + np.float32(0.75)
        new = new.reshape(sd[k].shape)
        tensors[k].data.view(-1)[flat] = float(new.reshape(-1)[flat])  # one element, through the alias no version counter sees
        sd2[k] = new
    got = model(_t(ev), _t(img), _t(mask))
    exp = build(sd2)(_t(ev), _t(img), _t(mask))
    for side in (0, 1):
        assert torch.equal(got[side]["sparse_positions"][0], exp[side]["sparse_positions"][0])
        assert torch.equal(got[side]["sparse_descriptors"][0], exp[side]["sparse_descriptors"][0])
        assert torch.equal(got[side]["raw_descriptors"], exp[side]["raw_descriptors"])
    assert torch.equal(got[2]["matches0"][0], exp[2]["matches0"][0])
    assert torch.equal(got[2]["matching_scores0"][0], exp[2]["matching_scores0"][0])
    # each edit alone is noticed too (the images are current again after the forward above)
    for k in edits:
        t = tensors[k].data.view(-1)
        flat = int(t.numel() // 2 + 1)
        t[flat] = t[flat] + 0.5
        a = model(_t(ev), _t(img), _t(mask))
        sd2[k] = _np(tensors[k].data).copy()
        e = build(sd2)(_t(ev), _t(img), _t(mask))
        assert torch.equal(a[0]["raw_descriptors"], e[0]["raw_descriptors"]) and torch.equal(a[1]["raw_descriptors"], e[1]["raw_descriptors"]), k
        assert torch.equal(a[2]["matching_scores0"][0], e[2]["matching_scores0"][0]), k


def test_forwards_on_other_caller_streams_share_the_process_wide_fork_streams():
    """Round 5: the small-batch head fork uses ONE library-owned side stream per (device, caller stream) for the whole process
    (HIP deals streams onto four compute pipes in creation order: per-handle fork streams created late landed on their callers'
    pipes).  Single pairs from two models, on the default stream and on a stream of the caller's own, prepared explicitly or
    not, give the results of the first call; a second prepare of the same stream is a no-op."""
    import ctypes
    from helpers import synth
    cfg = pkg.default_config("SP_MNN", event_channels=5)
    models = []
    for seed in (51, 51):
        m = pkg.EIM(cfg, device=DEV).eval()
        sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in m.state_dict().items()], seed=seed)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        models.append(m)
    ev, mask = synth.synth_events(97, 1, 5)
    img = synth.synth_image(97, 1)
    ref = models[0](_t(ev), _t(img.copy()), _t(mask))
    own = torch.cuda.Stream(DEV)
    L = pkg.native.lib()
    assert L.einx_fork_stream_prepare(ctypes.c_void_p(own.cuda_stream)) == 0
    assert L.einx_fork_stream_prepare(ctypes.c_void_p(own.cuda_stream)) == 0
    other = torch.cuda.Stream(DEV)  # not prepared: created at its first fork
    for m, stream in ((models[1], own), (models[0], other), (models[1], None)):
        e_, i_, k_ = _t(ev), _t(img.copy()), _t(mask)
        torch.cuda.synchronize()
        if stream is None:
            out = m(e_, i_, k_)
        else:
            with torch.cuda.stream(stream):
                out = m(e_, i_, k_)
            stream.synchronize()
        for a, b in zip(ref[:2], out[:2]):
            assert torch.equal(a["sparse_positions"][0], b["sparse_positions"][0])
            assert torch.equal(a["sparse_descriptors"][0], b["sparse_descriptors"][0])
        assert torch.equal(ref[2]["matches0"][0], out[2]["matches0"][0])


# ------------------------------------------------------------------ weight watch
def test_weight_watch_sees_permutations_and_sum_preserving_edits():
    """Round 5's watch summed ((position << 32) + word) * constant over a row: linear, so the hash depended on the SUM of a row's
    words only -- `p.data.copy_(p.data.flip(0))` on a bias / BatchNorm vector / the 576-word first convolution, or a +5 / -5 edit
    of two words' bit patterns, left native weight images stale.  The terms now go through a non-linear 64-bit finaliser
    (csrc/einx_common.h::einx_watch_term): each of those edits alone makes the next forward rebuild the images, and the
    result equals a model built from the edited weights."""
    cfg, model, sd = _eim_model("SP_MNN", 47)
    ev, mask = synth.synth_events(63, 1, 5)
    img = synth.synth_image(63, 1)
    run = lambda m: m(_t(ev), _t(img), _t(mask))  # noqa: E731
    run(model)
    tensors = dict(model.named_parameters())
    tensors.update(dict(model.named_buffers()))

    def fresh(sd_):
        m = pkg.EIM(cfg, device=DEV).eval()
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_.items()}, strict=False)
        return m

    def flip_whole(t):  # a permutation of the words of one watch row (all of these tensors are shorter than a 4096-word row)
        t.data.copy_(t.data.flatten().flip(0).view_as(t.data))

    def swap_two(t):  # two words trade places
        v = t.data.view(-1)
        a, b = v[1].item(), v[v.numel() // 2].item()
        v[1], v[v.numel() // 2] = b, a

    def plus_minus(t):  # +5 / -5 on the BIT PATTERNS of two words: the sum of the row's words is unchanged
        v = t.data.view(-1).view(torch.int32)
        v[2] += 5
        v[v.numel() - 3] -= 5

    edits = [("image_extractor.extractor.conv1a.weight", flip_whole), ("image_extractor.extractor.conv3b.bias", flip_whole),
             ("event_extractor.extractor.backbone.l2.1.2.running_var", swap_two), ("event_extractor.extractor.backbone.l1.0.0.bias", plus_minus),
             ("image_extractor.extractor.convDb.bias", plus_minus)]
    sd2 = dict(sd)
    for key, edit in edits:
        before = _np(tensors[key].data).copy()
        edit(tensors[key])
        after = _np(tensors[key].data).copy()
        assert not np.array_equal(before, after), key
        if edit is not plus_minus:
            assert np.array_equal(np.sort(before.reshape(-1)), np.sort(after.reshape(-1))), key  # a pure permutation
        else:
            assert int(before.view(np.int32).astype(np.int64).sum()) == int(after.view(np.int32).astype(np.int64).sum()), key
        sd2[key] = after
        got = run(model)
        exp = run(fresh(sd2))
        for side in (0, 1):
            assert torch.equal(got[side]["raw_descriptors"], exp[side]["raw_descriptors"]), key
            assert torch.equal(got[side]["logits"], exp[side]["logits"]), key
            assert torch.equal(got[side]["sparse_positions"][0], exp[side]["sparse_positions"][0]), key
        assert torch.equal(got[2]["matches0"][0], exp[2]["matches0"][0]), key


@pytest.mark.parametrize("name", list(RGB.cases))
def test_superpoint_takes_rgb_and_non_contiguous_images(oracle, name):
    """The reference: `image /= 255.0` (in place, whatever the strides) then `rgb_to_grayscale` for 3-channel images
    (superpoint_extractor.py:372-376).  Round 5 refused both.  Bit-equal to the oracle, equal to the reference's fixtures
    (tests/golden/rgb.npz), the caller's tensor is left scaled in place exactly as the reference leaves it, and the full model
    (EIM.forward, forward_graph) takes the same inputs."""
    from test_oracle_golden import _check_feats
    c = RGB.cases[name]
    cfg = pkg.configs.to_attr(c["cfg"])
    model = pkg.EIM(cfg, device=DEV)
    sd = state_dict_for(c, RGB)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model.eval()
    ext = model.image_extractor.extractor
    ext.dense_outputs = False
    x = rgb_input(c)
    ev, mk = synth.synth_events(c["iseed"], c["B"], 5, c["H"], c["W"])
    mask = mk if c["mask"] else None
    xt = _with_layout(x)
    assert xt.stride() == tuple(s // 4 for s in x.strides) and (c["layout"] == "rgb") == xt.is_contiguous()
    imf = ext(xt, None if mask is None else _t(mask))
    icfg = c["cfg"]["image_extractor"]["superpointv1"]
    xo = rgb_input(c)
    exp = oracle.extractor_forward("superpointv1", sub_dict(sd, "image_extractor.extractor."), xo, mask, top_k=icfg["detection_top_k"],
                                   radius=icfg["nms_radius"], border=icfg["remove_borders"], det_thr=icfg["detection_threshold"],
                                   scale=icfg["descriptor_scale_factor"])
    _feats_equal_oracle(imf, exp)
    as_np = {k: (_np(v) if torch.is_tensor(v) else [_np(t) for t in v]) for k, v in imf.items()}
    _check_feats(f"{name}.im", as_np, RGB)
    # the caller's tensor: scaled in place through its strides, still RGB / strided, equal to what the reference leaves behind
    after = _np(xt)
    assert np.array_equal(after, np.ascontiguousarray(xo))
    fx = RGB[f"{name}.after"]
    assert np.array_equal(after if fx.ndim == 4 else np.ascontiguousarray(after).reshape(-1)[::7], fx)
    # the whole model on the same layouts (eager and as a graph): same image-side features
    for fwd in (model.forward, model.forward_graph):
        xt2 = _with_layout(rgb_input(c))
        ef, imf2, m = fwd(_t(ev), xt2, _t(mk))
        for b in range(c["B"]):
            assert np.array_equal(_np(imf2["sparse_positions"][b]), exp["sparse_positions"][b]) or c["mask"]  # (EIM hands no image mask)
        if not c["mask"]:
            assert np.array_equal(_np(imf2["raw_descriptors"]), exp["raw_descriptors"])
        if fwd == model.forward:
            assert np.array_equal(_np(xt2), np.ascontiguousarray(xo))  # eager: the caller's tensor is scaled (forward_graph scales its copy)


def test_superpoint_wrong_channel_count_raises_like_conv1a():
    cfg, model, _ = _eim_model("SP_MNN", 5)
    with pytest.raises(RuntimeError, match="to have 1 channels, but got 2 channels instead"):
        model.image_extractor.extractor(torch.zeros(1, 2, 40, 48, device=DEV))


# ------------------------------------------------------------------ library-owned side streams are bounded
def test_fork_streams_stay_bounded_over_many_caller_streams():
    """einx_extract forks its descriptor branch onto a library-owned side stream per (device, caller stream).  Round 5 kept every
    side for the life of the process: a server that creates a stream per request grew HIP streams + events without bound.  Now at
    most EINX_FORK_STREAMS_MAX sides exist (least recently used first out, never one that a call is using), their streams are
    lent from a pool of 8 per device, and einx_fork_stream_release drops one explicitly.  64 short-lived caller streams: the count stays bounded, every result equals
    the first one bit for bit."""
    import ctypes
    L = pkg.native.lib()
    cap = 16  # EINX_FORK_STREAMS_MAX (include/einx.h)
    cfg, model, _ = _eim_model("SP_MNN", 51)
    ev, mask = synth.synth_events(65, 1, 5)
    img = synth.synth_image(65, 1)
    evt, mt, src = _t(ev), _t(mask), _t(img)
    ref = model(evt, src.clone(), mt)
    torch.cuda.synchronize()
    seen, lent = [], set()
    for i in range(64):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            got = model(evt, src.clone(), mt)
        s.synchronize()
        for side in (0, 1):
            assert torch.equal(got[side]["sparse_positions"][0], ref[side]["sparse_positions"][0]), i
            assert torch.equal(got[side]["sparse_descriptors"][0], ref[side]["sparse_descriptors"][0]), i
        assert torch.equal(got[2]["matches0"][0], ref[2]["matches0"][0]), i
        seen.append(L.einx_fork_stream_count())
        assert seen[-1] <= cap, seen
        lent.add(L.einx_fork_stream_of(ctypes.c_void_p(s.cuda_stream)))
        if i % 3 == 0:  # a host that tears its stream down tells the library
            before = L.einx_fork_stream_count()
            assert L.einx_fork_stream_release(ctypes.c_void_p(s.cuda_stream)) == 0
            assert L.einx_fork_stream_count() <= before
        del s
    assert max(seen) <= cap and L.einx_fork_stream_count() <= cap
    # the side streams are lent from a pool of EINX_FORK_STREAM_POOL (8) streams per device that is never destroyed (destroying
    # streams between hipGraph captures crashed hipGraphLaunch: profiles/r06_notes.md 7)
    assert None not in lent and 1 <= len(lent) <= 8, lent
    # releasing a stream that has no side is a no-op; the current stream's side comes back on demand
    assert L.einx_fork_stream_release(ctypes.c_void_p(12345)) == 0
    cur = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.einx_fork_stream_release(cur) == 0
    got = model(evt, src.clone(), mt)
    assert torch.equal(got[0]["sparse_descriptors"][0], ref[0]["sparse_descriptors"][0])


def test_side_streams_are_chosen_to_run_beside_their_callers():
    """HIP deals streams onto a few hardware queues; a side stream on its caller's queue serialises the fork.  Round 6 probes at
    creation (einx_stream_overlap_us: one wave spinning on each stream between a common start and end; elapsed / spin ~1.1 side
    by side, ~2.1 on one queue).  The probe itself: a stream against itself reads ~2, bad arguments are refused.  After a forward:
    the event extractor's side stream and both fork streams overlap with their callers."""
    import ctypes
    L = pkg.native.lib()
    EIM = import_module(pkg.__name__ + ".core.modules.EIM").EIM
    cfg, model, _ = _eim_model("SP_MNN", 52)
    ev, mask = synth.synth_events(65, 1, 5)
    model(_t(ev), _t(synth.synth_image(65, 1)), _t(mask))
    torch.cuda.synchronize()

    def ratio(a, b, spin=200):
        us = ctypes.c_float()
        assert L.einx_stream_overlap_us(ctypes.c_void_p(a), ctypes.c_void_p(b), spin, ctypes.byref(us)) == 0
        return us.value / spin

    cur = torch.cuda.current_stream().cuda_stream
    same = ratio(cur, cur)
    assert 1.7 < same < 2.6, same
    us = ctypes.c_float()
    assert L.einx_stream_overlap_us(ctypes.c_void_p(cur), ctypes.c_void_p(cur), 0, ctypes.byref(us)) != 0
    assert L.einx_stream_overlap_us(ctypes.c_void_p(cur), ctypes.c_void_p(cur), 100, None) != 0
    side = EIM._side_streams[("cuda", torch.cuda.current_device(), cur)].cuda_stream
    f_main, f_side = L.einx_fork_stream_of(ctypes.c_void_p(cur)), L.einx_fork_stream_of(ctypes.c_void_p(side))
    assert f_main and f_side and f_main != f_side
    assert L.einx_fork_stream_of(ctypes.c_void_p(f_main)) is None  # a side stream has no side of its own
    got = {"main|side": ratio(cur, side), "main|fork(main)": ratio(cur, f_main), "side|fork(side)": ratio(side, f_side)}
    assert all(r < 1.6 for r in got.values()), got
    # a forward on another caller stream gets a side stream of its own, beside THAT stream
    s2 = torch.cuda.Stream()
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        model(_t(ev), _t(synth.synth_image(65, 1)), _t(mask))
    s2.synchronize()
    side2 = EIM._side_streams[("cuda", torch.cuda.current_device(), s2.cuda_stream)].cuda_stream
    assert side2 != s2.cuda_stream and ratio(s2.cuda_stream, side2) < 1.6
    assert len(EIM._side_streams) <= EIM._SIDE_STREAMS_MAX


def test_silk_family_runs_on_one_stream_beyond_a_single_image():
    """The event extractor runs beside the image extractor on a second stream while the latency-bound tails are a visible share of
    the forward: always for the 1/8-resolution networks; the full-resolution networks only up to 2^17 pixels per call -- beyond that one
    stream is as fast or faster and the two-stream step was bimodal from process to process (profiles/r06_notes.md 9).  Either way
    the outputs are the same bits."""
    _, sp, _ = _eim_model("SP_MNN", 53)
    _, silk, _ = _eim_model("SiLK_MNN", 54)
    one = torch.zeros(1, 1, 260, 346, device=DEV)
    two = torch.zeros(2, 1, 260, 346, device=DEV)
    assert sp._two_streams_pay(one) and sp._two_streams_pay(torch.zeros(64, 1, 260, 346, device=DEV))
    assert silk._two_streams_pay(one) and not silk._two_streams_pay(two)
    ev, mask = synth.synth_events(66, 2, 5, 48, 56)
    img = synth.synth_image(67, 2, 48, 56)
    a = silk(_t(ev), _t(img.copy()), _t(mask))  # 2 x 48 x 56 pixels: two streams
    silk.overlap_extractors = False
    try:
        b = silk(_t(ev), _t(img.copy()), _t(mask))
    finally:
        del silk.overlap_extractors
    for side in (0, 1):
        for i in range(2):
            assert torch.equal(a[side]["sparse_positions"][i], b[side]["sparse_positions"][i])
            assert torch.equal(a[side]["sparse_descriptors"][i], b[side]["sparse_descriptors"][i])
    assert torch.equal(a[2]["matches0"][0], b[2]["matches0"][0])


def test_abi_version_and_struct_size_guards():
    """the public structs changed layout with no guard.  Now einx_abi_version() == EINX_ABI_VERSION, and a struct whose
    struct_size does not match the library's is refused with an error instead of being read as garbage."""
    import ctypes
    from importlib import import_module
    _lib = import_module(pkg.__name__ + "._lib")
    L = pkg.native.lib()
    assert L.einx_abi_version() == 6 and b"ABI 6" in L.einx_version()
    d = _lib.ExtractorDesc()
    d.struct_size = ctypes.sizeof(_lib.ExtractorDesc) - 8  # a host built against a shorter header
    assert not L.einx_extractor_create(ctypes.byref(d))
    assert b"struct_size" in L.einx_last_error()


# ------------------------------------------------------------------ the numeric contract on the device
def test_math_contract_device_bits_equal_the_oracles(oracle):
    """include/einx_math.h compiled by hipcc for gfx950 gives the SAME bits as compiled by gcc for the oracle, for every function
    and 200k arguments each (the header against float64 libm: tests/test_oracle_golden.py::test_math_contract_vs_libm, CPU suite)."""
    import ctypes
    from test_oracle_golden import _MATH, math_eval_inputs
    L = pkg.native.lib()
    O = oracle.lib()
    O.orc_math_eval.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
    for name, fn, _, rng in _MATH:
        x = math_eval_inputs(name, *rng)
        exp = np.empty_like(x)
        assert O.orc_math_eval(fn, x.ctypes.data, x.size, exp.ctypes.data) == 0
        xt = _t(x)
        yt = torch.empty_like(xt)
        assert L.einx_math_eval(fn, ctypes.c_void_p(xt.data_ptr()), x.size, ctypes.c_void_p(yt.data_ptr()), None) == 0
        torch.cuda.synchronize()
        got = _np(yt)
        same = got.view(np.uint32) == exp.view(np.uint32)
        assert same.all(), (name, x[~same][:5], got[~same][:5], exp[~same][:5])
