"""Round-3 GPU tests (-m gpu): maximum sizes.

One MI355X holds 288 GB, so a caller may hand the path batches whose activation buffers pass 2^31 ELEMENTS (the point where
a 32-bit element index wraps): B = 384 makes conv1a / conv1b write 384 * 64 * 264 * 352 = 2.28e9 floats into one buffer, and
B = 96 makes the dense descriptor map 96 * 256 * 260 * 346 = 2.21e9 floats.  The batch is built from four distinct pairs
repeated, so every pair of the large batch has a known bit-exact answer: the same pair run in a batch of four (which the
other tests pin to the oracle).
"""
import numpy as np
import pytest
import torch

from helpers import load_pkg, synth

pytestmark = pytest.mark.gpu
pkg = load_pkg()
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _require_gpu():
    assert torch.cuda.is_available(), "these tests need a HIP device"
    yield
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _model(dense_event=False):
    cfg = pkg.default_config("SP_MNN", event_channels=5)
    model = pkg.EIM(cfg, device=DEV).eval()
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=11)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model.event_extractor.extractor.dense_outputs = dense_event
    model.image_extractor.extractor.dense_outputs = False
    return model


def _four_pairs(seed):
    ev, mask = synth.synth_events(seed, 4, 5)
    img = synth.synth_image(seed, 4)
    return ev, mask, img


def _tiled(a, B):
    return np.concatenate([a] * (B // a.shape[0]), axis=0)


def _need_free_gb(gb):
    free = torch.cuda.mem_get_info()[0] / 2**30
    if free < gb:
        pytest.skip(f"needs {gb} GB of free device memory, {free:.0f} GB available")


def test_batch_384_activations_beyond_2_31_elements_equal_the_small_batch():
    _need_free_gb(96)
    B = 384
    assert B * 64 * 264 * 352 > 2**31
    model = _model()
    ev, mask, img = _four_pairs(4242)
    ef4, if4, m4 = model(_t(ev), _t(img), _t(mask))
    ef, imf, m = model(_t(_tiled(ev, B)), _t(_tiled(img, B)), _t(_tiled(mask, B)))
    assert len(ef["sparse_positions"]) == B and len(m["matches0"]) == B
    for b in (0, 1, 2, 3, 189, 190, 191, 192, 193, 362, 363, 380, 381, 382, 383):  # 362 is the first image past 2^31 floats of conv1a
        r = b % 4
        for got, exp in ((ef, ef4), (imf, if4)):
            assert torch.equal(got["sparse_positions"][b], exp["sparse_positions"][r]), f"pair {b}: keypoints"
            assert torch.equal(got["sparse_descriptors"][b], exp["sparse_descriptors"][r]), f"pair {b}: descriptors"
            assert torch.equal(got["score"][b], exp["score"][r]), f"pair {b}: score map"
            assert torch.equal(got["coarse_descriptors"][b], exp["coarse_descriptors"][r]), f"pair {b}: coarse descriptors"
        for k in ("matches0", "matches1", "matching_scores0", "matched_kpts0", "matched_kpts1", "log_assignment"):
            assert torch.equal(m[k][b], m4[k][r]), f"pair {b}: {k}"
    assert int(ef["sparse_positions"][383].shape[0]) > 100


def test_batch_96_dense_descriptor_map_beyond_2_31_elements_equals_the_small_batch():
    _need_free_gb(48)
    B = 96
    assert B * 256 * 260 * 346 > 2**31
    model = _model(dense_event=True)
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


@pytest.mark.parametrize("n", [1, 255, 1023, 1024, 1025, 4097, 260 * 346, 3 * 260 * 346 + 5])
def test_div_inplace_sizes(n):
    N = pkg.native
    x = torch.arange(n, dtype=torch.float32, device=DEV) * 0.37 + 1.0
    ref = (x.cpu().numpy() / np.float32(255.0)).astype(np.float32)
    N.div_inplace(x, 255.0)
    assert np.array_equal(x.cpu().numpy(), ref)


# ------------------------------------------------------------------ EIM.forward's optional masks (EIM.py:44: events_mask=None, image_mask=None)
def test_eim_forward_without_masks_and_with_an_image_mask_vs_oracle(oracle):
    from helpers import sub_dict
    cfg = pkg.default_config("SP_MNN", event_channels=5)
    for sec in (cfg.event_extractor.vgg, cfg.image_extractor.superpointv1):
        sec.detection_top_k = 150
    model = pkg.EIM(cfg, device=DEV).eval()
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=17)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    for ext in (model.event_extractor.extractor, model.image_extractor.extractor):
        ext.dense_outputs = False
    H, W, B = 84, 108, 2
    ev, emask = synth.synth_events(91, B, 5, H, W)
    img = synth.synth_image(91, B, H, W)
    imask = synth.uniform01(92, (B, 1, H, W)) < np.float32(0.6)
    esub, isub = sub_dict(sd, "event_extractor.extractor."), sub_dict(sd, "image_extractor.extractor.")
    for em, im in ((None, None), (emask, imask), (None, imask)):
        ef, imf, m = model(_t(ev), _t(img), None if em is None else _t(em), None if im is None else _t(im))
        oe = oracle.extractor_forward("vgg", esub, ev.copy(), em, top_k=150)
        oi = oracle.extractor_forward("superpointv1", isub, img.copy(), im, top_k=150)
        for got, exp in ((ef, oe), (imf, oi)):
            assert np.array_equal(got["score"].cpu().numpy(), exp["score"])
            assert np.array_equal(got["nms"].cpu().numpy(), exp["nms"])
            for b in range(B):
                assert np.array_equal(got["sparse_positions"][b].cpu().numpy(), exp["sparse_positions"][b])
                assert np.array_equal(got["sparse_descriptors"][b].cpu().numpy(), exp["sparse_descriptors"][b])
        for b in range(B):
            r = oracle.mnn(oe["sparse_descriptors"][b], oi["sparse_descriptors"][b], want_la=False)
            assert np.array_equal(m["matches0"][b].cpu().numpy()[0], r["matches0"])
        if im is not None:  # the image-side mask zeroes scores exactly where it is False (no dilation on the image side)
            assert float(imf["score"][~_t(im)].abs().max()) == 0.0


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


# ------------------------------------------------------------------ whole forwards at odd geometries (small-grid kernels, forked heads, ragged tiles)
@pytest.mark.parametrize("cfg_name,B,H,W,bins,k", [
    ("SP_MNN", 1, 41, 67, 5, 64), ("SP_MNN", 2, 97, 53, 3, 200), ("SP_MNN", 3, 64, 136, 16, 128), ("SP_MNN", 1, 135, 181, 5, 500),
    ("SiLK_MNN", 1, 37, 45, 5, 80), ("SiLK_MNN", 2, 58, 83, 2, 150),
], ids=lambda v: str(v))
def test_whole_forward_at_odd_geometries_vs_oracle(oracle, cfg_name, B, H, W, bins, k):
    """Single pairs and tiny batches at sizes that are not multiples of the cell / tile sizes: every launch takes a small-grid
    path (conv16_kernel, conv16_1x1_kernel, forked head branches) with ragged tiles; keypoints, descriptors and matches must
    equal per-pair oracle runs bit for bit."""
    from helpers import sub_dict
    cfg = pkg.default_config(cfg_name, event_channels=bins)
    et, it = cfg.event_extractor.type, cfg.image_extractor.type
    cfg.event_extractor[et].detection_top_k = k
    cfg.image_extractor[it].detection_top_k = k
    model = pkg.EIM(cfg, device=DEV).eval()
    sd = synth.synth_state_dict([(kk, tuple(v.shape)) for kk, v in model.state_dict().items()], seed=H * 7 + W)
    model.load_state_dict({kk: torch.from_numpy(v) for kk, v in sd.items()}, strict=False)
    for ext in (model.event_extractor.extractor, model.image_extractor.extractor):
        ext.dense_outputs = False
    ev, mask = synth.synth_events(H + W, B, bins, H, W)
    img = synth.synth_image(H + W, B, H, W)
    ef, imf, m = model(_t(ev), _t(img), _t(mask))
    oe = oracle.extractor_forward(et, sub_dict(sd, "event_extractor.extractor."), ev.copy(), mask, top_k=k,
                                  scale=cfg.event_extractor[et].descriptor_scale_factor)
    oi = oracle.extractor_forward(it, sub_dict(sd, "image_extractor.extractor."), img.copy(), None, top_k=k,
                                  scale=cfg.image_extractor[it].descriptor_scale_factor)
    for got, exp in ((ef, oe), (imf, oi)):
        for key in ("backbone_feats", "logits", "raw_descriptors", "score", "nms"):
            assert np.array_equal(got[key].cpu().numpy(), exp[key]), key
        for b in range(B):
            assert np.array_equal(got["sparse_positions"][b].cpu().numpy(), exp["sparse_positions"][b])
            assert np.array_equal(got["sparse_descriptors"][b].cpu().numpy(), exp["sparse_descriptors"][b])
    for b in range(B):
        if len(oe["sparse_descriptors"][b]) == 0 or len(oi["sparse_descriptors"][b]) == 0:
            continue
        r = oracle.mnn(oe["sparse_descriptors"][b], oi["sparse_descriptors"][b], want_la=False)
        assert np.array_equal(m["matches0"][b].cpu().numpy()[0], r["matches0"])


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
