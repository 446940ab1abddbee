"""GPU tests (-m gpu), component: conv.
SURVEY 8a rows A1 (Padder fold), A3-A6 (VGG / SuperPoint / SiLK convolution stacks): conv.hip against the oracle, bit for bit.
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

from helpers import (score_map, sub_dict, synth)
from gpu_support import (CONV16_1X1_SHAPES, CONV16_SHAPES, CONV_SEEDS, CONV_SHAPES, DEV, _four_pairs, _need_free_gb, _np, _rng,
                         _sp_mnn_model, _t, _tiled, pkg)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", CONV_SHAPES, ids=lambda s: "x".join(str(v) for v in s[:8]))
def test_conv_block_bit_exact(oracle, shape):
    cin, cout, H, W, ks, relu, bn, pool, fold = shape
    seed = 1000 + cin * 7 + cout
    B = 2
    Hs, Ws = (fold[2], fold[3]) if fold else (H, W)
    x = synth.normalish(seed, (B, cin, Hs, Ws))
    w = synth.synth_param("c.weight", (cout, cin, ks, ks), seed)
    b = synth.uniform(seed + 1, (cout,), -0.5, 0.5)
    bnp = None
    scale = shift = None
    if bn:
        g, be = synth.uniform(seed + 2, (cout,), 0.5, 1.5), synth.uniform(seed + 3, (cout,), -0.3, 0.3)
        mu, var = synth.uniform(seed + 4, (cout,), -0.3, 0.3), synth.uniform(seed + 5, (cout,), 0.5, 1.5)
        g[0] = -g[0]  # negative gain: BN must stay after ReLU and before the pool
        scale, shift = oracle.bn_fold(g, be, mu, var)
        bnp = (_t(g), _t(be), _t(mu), _t(var), 1e-5)
    xin = x
    if fold:
        h0, w0 = fold[0], fold[1]
        xin = oracle.pad_replicate(x, (w0, W - Ws - w0, h0, H - Hs - h0))
    exp = oracle.conv_block(xin, w, b, scale, shift, relu=relu, pool=pool)
    layer = pkg.native.ConvLayer(_t(w), _t(b), bnp, relu=relu, pool=pool)
    got = layer(_t(x), fold=(fold[0], fold[1], H, W) if fold else None)
    assert np.array_equal(_np(got), exp)


@pytest.mark.parametrize("pool", [True, False])
def test_conv_block_three_per_cu_instantiation_bit_exact(oracle, pool):
    """Round 5: launches of at least eight rounds of the 8x32 tiles (>= 6144 workgroups) take the instantiation that fits three
    workgroups per CU (<= 80 registers; the un-pooled one spills three dwords).  256 small images reach that count."""
    B, cin, cout, H, W = 256, 8, 64, 64, 96
    x = synth.normalish(77, (B, cin, H, W))
    w = synth.synth_param("c.weight", (cout, cin, 3, 3), 78)
    b = synth.uniform(79, (cout,), -0.5, 0.5)
    g, be = synth.uniform(80, (cout,), 0.5, 1.5), synth.uniform(81, (cout,), -0.3, 0.3)
    mu, var = synth.uniform(82, (cout,), -0.3, 0.3), synth.uniform(83, (cout,), 0.5, 1.5)
    g[3] = -g[3]
    scale, shift = oracle.bn_fold(g, be, mu, var)
    exp = oracle.conv_block(x, w, b, scale, shift, relu=True, pool=pool)
    layer = pkg.native.ConvLayer(_t(w), _t(b), (_t(g), _t(be), _t(mu), _t(var), 1e-5), relu=True, pool=pool)
    got = layer(_t(x))
    name = pkg.native.lib().einx_conv_last_kernel().decode()
    assert name == f"conv_block_kernel<3,8,32,2,4,1,2,8,{'true' if pool else 'false'}> (3 per CU)", name
    assert np.array_equal(_np(got), exp)


@pytest.mark.parametrize("case", [dict(B=18, H=260, W=346, fold=(2, 3, 264, 352), pool=True, bn0=False, relu1=True),   # SuperPointv1 at 346x260 (replicate fold)
                                  dict(B=640, H=40, W=48, fold=None, pool=True, bn0=True, relu1=True),               # small maps, BatchNorm after the first layer
                                  dict(B=420, H=37, W=77, fold=(1, 1, 40, 80), pool=False, bn0=True, relu1=False),   # un-pooled second layer, partial tiles
                                  dict(B=1100, H=24, W=40, fold=None, pool=True, bn0=False, relu1=True),
                                  dict(B=18, H=260, W=346, fold=(2, 3, 264, 352), pool=True, bn0=True, relu1=True, cin=5, bn1=True),  # VGGExtractor, 5 event bins
                                  dict(B=420, H=37, W=77, fold=(1, 1, 40, 80), pool=False, bn0=True, relu1=True, cin=5, bn1=True),
                                  dict(B=1100, H=24, W=40, fold=None, pool=True, bn0=False, relu1=False, cin=5, bn1=False)],
                         ids=["sp_346x260", "small_bn", "unpooled_partial", "tiny", "vgg5_346x260", "vgg5_unpooled_partial", "vgg5_tiny"])
def test_first_two_layers_fused_bit_exact(oracle, case):
    """Round 6: the thin first layer (1 -> 64: SuperPointv1; 5 -> 64: the event extractor at BASELINE's 5 bins) recomputed inside the
    second layer's launch on the matrix cores (conv1ab_kernel<CIN0>): the output of the second layer is bit-equal to the two launches
    it replaces (every image) and to the oracle's two conv blocks (sampled images)."""
    import ctypes
    B, H, W, fold = case["B"], case["H"], case["W"], case["fold"]
    cin = case.get("cin", 1)
    x = synth.normalish(91, (B, cin, H, W))
    w0 = synth.synth_param("a.weight", (64, cin, 3, 3), 92)
    b0 = synth.uniform(93, (64,), -0.5, 0.5)
    w1 = synth.synth_param("b.weight", (64, 64, 3, 3), 94)
    b1 = synth.uniform(95, (64,), -0.5, 0.5)
    bn0 = s0 = t0 = None
    if case["bn0"]:
        g, be = synth.uniform(96, (64,), 0.5, 1.5), synth.uniform(97, (64,), -0.3, 0.3)
        mu, var = synth.uniform(98, (64,), -0.3, 0.3), synth.uniform(99, (64,), 0.5, 1.5)
        g[5] = -g[5]
        s0, t0 = oracle.bn_fold(g, be, mu, var)
        bn0 = (_t(g), _t(be), _t(mu), _t(var), 1e-5)
    l0 = pkg.native.ConvLayer(_t(w0), _t(b0), bn0, relu=True, pool=False)
    bn1 = s1 = t1 = None
    if case.get("bn1"):
        g1, be1 = synth.uniform(196, (64,), 0.5, 1.5), synth.uniform(197, (64,), -0.3, 0.3)
        mu1, var1 = synth.uniform(198, (64,), -0.3, 0.3), synth.uniform(199, (64,), 0.5, 1.5)
        g1[9] = -g1[9]
        s1, t1 = oracle.bn_fold(g1, be1, mu1, var1)
        bn1 = (_t(g1), _t(be1), _t(mu1), _t(var1), 1e-5)
    l1 = pkg.native.ConvLayer(_t(w1), _t(b1), bn1, relu=case["relu1"], pool=case["pool"])
    xt = _t(x)
    Hp, Wp = (fold[2], fold[3]) if fold else (H, W)
    h0, w0_ = (fold[0], fold[1]) if fold else (0, 0)
    L = pkg.native.lib()
    # 1: the dispatcher fuses (one input channel); 2: covered and bit-exact but slower than two launches (five channels), never dispatched
    assert L.einx_conv_first_two_fused_ok(ctypes.byref(l0.desc), ctypes.byref(l1.desc), B, Hp, Wp) == (1 if cin == 1 else 2)
    assert L.einx_conv_first_two_fused_ok(ctypes.byref(l0.desc), ctypes.byref(l1.desc), 1, Hp, Wp) == 0  # small launches keep the two layers
    seq = l1(l0(xt, fold=(h0, w0_, Hp, Wp) if fold else None))
    Ho, Wo = (Hp // 2, Wp // 2) if case["pool"] else (Hp, Wp)
    got = torch.full((B, 64, Ho, Wo), float("nan"), dtype=torch.float32, device=DEV)
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    rc = L.einx_conv_first_two_fused(P(xt), B, H, W, h0, w0_, Hp, Wp, ctypes.byref(l0.desc), ctypes.byref(l1.desc), P(got), None)
    assert rc == 0, L.einx_last_error()
    assert L.einx_conv_last_kernel().decode().startswith("conv1ab_kernel")
    torch.cuda.synchronize()
    assert torch.equal(got, seq)
    for b in (0, B // 2, B - 1):  # the oracle on sampled images
        xb = x[b:b + 1]
        if fold:
            pads = (w0_, Wp - W - w0_, h0, Hp - H - h0)
            xb = oracle.pad_replicate(xb, pads)
        mid = oracle.conv_block(xb, w0, b0, s0, t0, relu=True, pool=False)
        exp = oracle.conv_block(mid, w1, b1, s1, t1, relu=case["relu1"], pool=case["pool"])
        assert np.array_equal(_np(got[b:b + 1]), exp), b


@pytest.mark.parametrize("shape", CONV16_SHAPES, ids=lambda s: "x".join(str(v) for v in s))
def test_conv16_small_grid_kernel_bit_exact(oracle, shape):
    """conv16_kernel (v_mfma_f32_16x16x4_f32, one accumulator chain per output in the same K order as conv_block_kernel):
    bit-equal to the oracle, and actually the kernel the dispatcher launches for these small grids."""
    B, cin, cout, H, W, relu, bn, pool, npw = shape
    seed = 4000 + cin * 3 + cout + H
    x = synth.normalish(seed, (B, cin, H, W))
    w = synth.synth_param("c.weight", (cout, cin, 3, 3), seed)
    b = synth.uniform(seed + 1, (cout,), -0.5, 0.5)
    bnp = scale = shift = None
    if bn:
        g, be = synth.uniform(seed + 2, (cout,), 0.5, 1.5), synth.uniform(seed + 3, (cout,), -0.3, 0.3)
        mu, var = synth.uniform(seed + 4, (cout,), -0.3, 0.3), synth.uniform(seed + 5, (cout,), 0.5, 1.5)
        g[0] = -g[0]
        scale, shift = oracle.bn_fold(g, be, mu, var)
        bnp = (_t(g), _t(be), _t(mu), _t(var), 1e-5)
    exp = oracle.conv_block(x, w, b, scale, shift, relu=relu, pool=pool)
    layer = pkg.native.ConvLayer(_t(w), _t(b), bnp, relu=relu, pool=pool)
    got = layer(_t(x))
    name = pkg.native.lib().einx_conv_last_kernel().decode()
    assert name == f"conv16_kernel<{'true' if pool else 'false'},8,{npw}>", name
    assert np.array_equal(_np(got), exp)


@pytest.mark.parametrize("shape", CONV16_1X1_SHAPES, ids=lambda s: "x".join(str(v) for v in s))
def test_conv16_1x1_small_grid_kernel_bit_exact(oracle, shape):
    """conv16_1x1_kernel (1x1 layers of small grids on v_mfma_f32_16x16x4_f32, K = input channels in natural order): bit-equal to
    the oracle and to conv_block_kernel<1,...> (the same layer at a batch the dispatcher sends there), and the kernel
    the dispatcher really launches."""
    B, cin, cout, H, W, relu, bn, npw = shape
    seed = 5000 + cin * 3 + cout + H
    x = synth.normalish(seed, (B, cin, H, W))
    w = synth.synth_param("c.weight", (cout, cin, 1, 1), seed)
    b = synth.uniform(seed + 1, (cout,), -0.5, 0.5)
    bnp = scale = shift = None
    if bn:
        g, be = synth.uniform(seed + 2, (cout,), 0.5, 1.5), synth.uniform(seed + 3, (cout,), -0.3, 0.3)
        mu, var = synth.uniform(seed + 4, (cout,), -0.3, 0.3), synth.uniform(seed + 5, (cout,), 0.5, 1.5)
        g[0] = -g[0]
        scale, shift = oracle.bn_fold(g, be, mu, var)
        bnp = (_t(g), _t(be), _t(mu), _t(var), 1e-5)
    exp = oracle.conv_block(x, w, b, scale, shift, relu=relu, pool=False)
    layer = pkg.native.ConvLayer(_t(w), _t(b), bnp, relu=relu, pool=False)
    got = layer(_t(x))
    name = pkg.native.lib().einx_conv_last_kernel().decode()
    assert name == f"conv16_1x1_kernel<{npw}>", name
    assert np.array_equal(_np(got), exp)
    # the large-grid kernel on the same images repeated: every copy equals the small-grid result
    reps = -(-600 * 128 // (H * W * ((cout + 63) // 64)))  # enough copies for >= 512 128-pixel workgroups
    if reps * B * cin * H * W <= 64 << 20:
        big = layer(_t(np.concatenate([x] * reps, 0)))
        assert pkg.native.lib().einx_conv_last_kernel().decode().startswith("conv_block_kernel<1,")
        assert torch.equal(big[:B], got) and torch.equal(big[-B:], got)


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


def test_batch_384_activations_beyond_2_31_elements_equal_the_small_batch():
    _need_free_gb(96)
    B = 384
    assert B * 64 * 264 * 352 > 2**31
    model = _sp_mnn_model()
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


@pytest.mark.parametrize("n", [1, 255, 1023, 1024, 1025, 4097, 260 * 346, 3 * 260 * 346 + 5])
def test_div_inplace_sizes(n):
    N = pkg.native
    x = torch.arange(n, dtype=torch.float32, device=DEV) * 0.37 + 1.0
    ref = (x.cpu().numpy() / np.float32(255.0)).astype(np.float32)
    N.div_inplace(x, 255.0)
    assert np.array_equal(x.cpu().numpy(), ref)


@pytest.mark.parametrize("cfg_name", ["SP_MNN", "SiLK_MNN"])
def test_single_image_merged_head_layer_equals_the_batched_path(oracle, cfg_name):
    """Single images run the two heads' first 3x3 layers as ONE launch (einx_extractor_desc::merged_head0: the detector's output
    channels, then the descriptor's); batches keep the two launches.  Same bits either way: image 0 alone == image 0 of a batch of
    three, for both extractor families (VGG heads with BatchNorm, SuperPoint's and SiLK's without), and == the oracle."""
    from helpers import sub_dict, synth
    cfg = pkg.default_config(cfg_name, event_channels=5)
    model = pkg.EIM(cfg, device=DEV).eval()
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=53)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    H, W = (120, 152) if cfg_name == "SP_MNN" else (64, 80)
    ev, mask = synth.synth_events(53, 3, 5, H, W)
    img = synth.synth_image(53, 3, H, W)
    for ext in (model.event_extractor.extractor, model.image_extractor.extractor):
        eng = ext.engine()
        assert eng.merged_head0 is not None and eng.merged_head0.cout == eng.det_head[0].cout + eng.desc_head[0].cout
    one = model(_t(ev[:1]), _t(img[:1].copy()), _t(mask[:1]))
    three = model(_t(ev), _t(img.copy()), _t(mask))
    for side in (0, 1):
        for key in ("logits", "raw_descriptors", "score"):
            assert torch.equal(one[side][key][0], three[side][key][0]), (side, key)
        assert torch.equal(one[side]["sparse_positions"][0], three[side]["sparse_positions"][0])
        assert torch.equal(one[side]["sparse_descriptors"][0], three[side]["sparse_descriptors"][0])
    kinds = ("vgg", "superpointv1") if cfg_name == "SP_MNN" else ("vgg_np", "silk")
    es, is_ = (float(e.descriptor_scale_factor.detach()) for e in (model.event_extractor.extractor, model.image_extractor.extractor))
    oe = oracle.extractor_forward(kinds[0], sub_dict(sd, "event_extractor.extractor."), ev[:1].copy(), mask[:1], top_k=1024, scale=es)
    oi = oracle.extractor_forward(kinds[1], sub_dict(sd, "image_extractor.extractor."), img[:1].copy(), None, top_k=1024, scale=is_)
    for got, exp in ((one[0], oe), (one[1], oi)):
        assert np.array_equal(_np(got["logits"]), exp["logits"]) and np.array_equal(_np(got["raw_descriptors"]), exp["raw_descriptors"])


@pytest.mark.parametrize("seed", CONV_SEEDS)
def test_random_conv_blocks(oracle, seed):
    r = _rng(1000 + seed)
    ks = int(r.choice([1, 3, 3, 3]))
    cin = int(r.choice([1, 2, 3, 5, 6, 7, 16, 33, 64, 128, 130]))
    cout = int(r.choice([1, 7, 64, 65, 96, 128, 200]))
    pool = bool(ks == 3 and r.random() < 0.4)
    H = int(r.integers(3, 70))
    W = int(r.integers(3, 90))
    if pool:
        H, W = H + (H & 1), W + (W & 1)
    B = int(r.integers(1, 4))
    relu, bn = bool(r.random() < 0.7), bool(r.random() < 0.5)
    fold = None
    if ks == 3 and r.random() < 0.3:  # replicate padding folded into the layer (first layers)
        h0, w0 = int(r.integers(0, 3)), int(r.integers(0, 4))
        Hs, Ws = H - h0 - int(r.integers(0, 3)), W - w0 - int(r.integers(0, 4))
        if Hs >= 1 and Ws >= 1:
            fold = (h0, w0, Hs, Ws)
    Hs, Ws = (fold[2], fold[3]) if fold else (H, W)
    x = synth.normalish(5000 + seed, (B, cin, Hs, Ws))
    if r.random() < 0.5:
        x = np.maximum(x, 0)  # ReLU-sparse inputs as inside the networks
    w = synth.synth_param("c.weight", (cout, cin, ks, ks), 6000 + seed)
    b = synth.uniform(7000 + seed, (cout,), -0.5, 0.5)
    scale = shift = bnp = None
    if bn:
        g, be = synth.uniform(seed + 2, (cout,), 0.5, 1.5), synth.uniform(seed + 3, (cout,), -0.3, 0.3)
        mu, var = synth.uniform(seed + 4, (cout,), -0.3, 0.3), synth.uniform(seed + 5, (cout,), 0.5, 1.5)
        scale, shift = oracle.bn_fold(g, be, mu, var)
        bnp = (_t(g), _t(be), _t(mu), _t(var), 1e-5)
    xin = x
    if fold:
        xin = oracle.pad_replicate(x, (fold[1], W - Ws - fold[1], fold[0], H - Hs - fold[0]))
    exp = oracle.conv_block(xin, w, b, scale, shift, relu=relu, pool=pool)
    layer = pkg.native.ConvLayer(_t(w), _t(b), bnp, relu=relu, pool=pool)
    got = layer(_t(x), fold=(fold[0], fold[1], H, W) if fold else None)
    assert np.array_equal(_np(got), exp), (ks, cin, cout, H, W, B, pool, relu, bn, fold)
