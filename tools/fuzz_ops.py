#!/usr/bin/env python3
"""Randomised differential campaign, op level (companion of fuzz_parity.py): random convolution layers (channels, sizes, kernel,
ReLU / BatchNorm / pool, replicate-pad fold), detection on tie-heavy score maps, dense descriptor upsampling, descriptor sampling,
MNN with ratio / distance thresholds on ragged counts, and the evaluation metrics -- GPU vs the CPU oracle, bit for bit (metrics:
1e-6).  One integer seed per case, printed on failure.      python tools/fuzz_ops.py [--seconds 300] [--seed0 1]
(test infrastructure: the oracle is the checker, never the product)"""
import argparse, importlib, os, sys, time, traceback
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ei-nexus_official_amd")
from oracle import oracle as orc  # noqa: E402
N = pkg.native
DEV = "cuda:0"
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
n = lambda x: x.detach().cpu().numpy()  # noqa: E731


def f32(r, shape, lo=-1.0, hi=1.0):
    return r.uniform(lo, hi, shape).astype(np.float32)


def conv_case(seed):
    r = np.random.default_rng(seed)
    ks = int(r.choice([1, 3, 3, 3]))
    cin = int(r.choice([1, 2, 3, 5, 6, 8, 16, 24, 64, 72, 128]))
    cout = int(r.choice([1, 5, 16, 64, 65, 70, 128, 130, 256]))
    B = int(r.choice([1, 2, 3]))
    H, W = int(r.integers(2, 70)), int(r.integers(2, 90))
    pool = bool(r.integers(2)) and ks == 3
    if pool:
        H, W = H + H % 2, W + W % 2
    relu, bn = bool(r.integers(2)), bool(r.integers(2))
    fold = None
    if ks == 3 and r.integers(4) == 0:
        h0, w0, h1, w1 = (int(v) for v in r.integers(0, 5, 4))
        if pool:
            h1 += (h0 + h1) % 2
            w1 += (w0 + w1) % 2
        fold = (h0, w0, H + h0 + h1, W + w0 + w1)
    desc = f"seed {seed}: conv ks={ks} {cin}->{cout} B={B} {H}x{W} pool={pool} relu={relu} bn={bn} fold={fold}"
    x = f32(r, (B, cin, H, W), -2, 2)
    if r.integers(3) == 0:
        x[x < 0] = 0  # sparse (post-ReLU-like) inputs
    w = f32(r, (cout, cin, ks, ks)) / np.float32(np.sqrt(cin * ks * ks))
    b = f32(r, (cout,), -0.5, 0.5)
    scale = shift = bnp = None
    if bn:
        g, be, mu, var = f32(r, (cout,), -1.5, 1.5), f32(r, (cout,), -0.3, 0.3), f32(r, (cout,), -0.3, 0.3), f32(r, (cout,), 0.5, 1.5)
        scale, shift = orc.bn_fold(g, be, mu, var)
        bnp = (t(g), t(be), t(mu), t(var), 1e-5)
    xin = x
    if fold:
        h0, w0, Hf, Wf = fold
        xin = orc.pad_replicate(x, (w0, Wf - W - w0, h0, Hf - H - h0))
    exp = orc.conv_block(xin, w, b, scale, shift, relu=relu, pool=pool)
    got = n(N.ConvLayer(t(w), t(b), bnp, relu=relu, pool=pool)(t(x), fold=fold))
    if got.shape != exp.shape or not np.array_equal(got, exp):
        raise AssertionError(f"{desc}: differs ({N.lib().einx_conv_last_kernel().decode()})")


def detect_case(seed):
    r = np.random.default_rng(seed)
    B = int(r.choice([1, 2, 4]))
    Hp, Wp = int(r.integers(8, 150)), int(r.integers(8, 200))
    pads = tuple(int(v) for v in (r.integers(0, 4), r.integers(0, 4), r.integers(0, 4), r.integers(0, 4)))  # (w0, w1, h0, h1)
    if Hp - pads[2] - pads[3] < 2 or Wp - pads[0] - pads[1] < 2:
        pads = (0, 0, 0, 0)
    radius = int(r.choice([0, 1, 2, 3, 4, 4]))
    border = int(r.choice([0, 2, 4]))
    top_k = int(r.choice([1, 5, 60, 1000]))
    levels = int(r.choice([2, 4, 16, 0]))  # quantised maps: long tie chains
    s = r.uniform(0, 1, (B, 1, Hp, Wp)).astype(np.float32) ** 3
    if levels:
        s = (np.floor(s * levels) / np.float32(levels)).astype(np.float32)
    desc = f"seed {seed}: detect B={B} {Hp}x{Wp} pads={pads} r={radius} border={border} top_k={top_k} levels={levels}"
    sc = s.copy()
    orc.mask_border(sc, None, pads, False, border)
    exp_nms, exp_pos, exp_idx, exp_thr, _ = orc.detect_post(sc.copy(), top_k, radius, border, 1.0, pads, "yx")
    d = N.detect(t(sc), top_k=top_k, radius=radius, det_thr=1.0, pads=pads)
    for _ in range(10):
        if not n(d.not_converged).any():
            break
        d = N.detect(t(sc), top_k=top_k, radius=radius, det_thr=1.0, pads=pads, nms_iters=4096)
    cnt = n(d.counts)
    if cnt.tolist() != [len(p) for p in exp_pos]:
        raise AssertionError(f"{desc}: counts {cnt.tolist()} vs {[len(p) for p in exp_pos]}")
    for b in range(B):
        if not (np.array_equal(n(d.positions[b, :cnt[b]]), exp_pos[b]) and np.array_equal(n(d.indices[b, :cnt[b]]), exp_idx[b])):
            raise AssertionError(f"{desc}: positions / indices of image {b} differ")
    if not np.array_equal(n(d.thr), exp_thr):
        raise AssertionError(f"{desc}: thresholds differ")


def upsample_case(seed):
    r = np.random.default_rng(seed)
    B, D = int(r.choice([1, 2])), int(r.choice([1, 7, 32, 33, 128, 256]))
    hc, wc = int(r.integers(2, 20)), int(r.integers(2, 30))
    f = int(r.choice([1, 2, 8, 8]))
    Hp, Wp = hc * f + int(r.integers(0, f)), wc * f + int(r.integers(0, f))
    h0, w0 = int(r.integers(0, min(4, Hp - 1))), int(r.integers(0, min(4, Wp - 1)))
    H, W = int(r.integers(1, Hp - h0 + 1)), int(r.integers(1, Wp - w0 + 1))
    sc = float(r.choice([1.0, 1.25, 0.5]))
    desc = f"seed {seed}: upsample B={B} D={D} {hc}x{wc} -> {Hp}x{Wp} crop ({h0},{w0},{H},{W}) scale={sc}"
    raw = f32(r, (B, D, hc, wc), -2, 2)
    got = n(N.upsample_normalize(t(raw), (Hp, Wp), (w0, Wp - w0 - W, h0, Hp - h0 - H), sc))
    exp = orc.upsample_normalize(raw, (Hp, Wp), sc)[:, :, h0:h0 + H, w0:w0 + W]
    if got.shape != exp.shape or not np.array_equal(got, exp, equal_nan=True):
        raise AssertionError(f"{desc}: differs")


def mnn_case(seed):
    r = np.random.default_rng(seed)
    D = int(r.choice([32, 64, 128, 256]))
    n0, n1 = int(r.choice([1, 2, 3, 17, 130, 700, 1024])), int(r.choice([2, 3, 17, 129, 700, 1024]))
    ratio = r.choice([None, 0.8, 0.95]) if n1 >= 2 and n0 >= 2 else None
    dist = r.choice([None, 0.7, 1.2])
    ratio = None if ratio is None else float(ratio)
    dist = None if dist is None else float(dist)
    desc = f"seed {seed}: mnn D={D} n0={n0} n1={n1} ratio={ratio} dist={dist}"
    d0 = f32(r, (n0, D))
    d1 = f32(r, (n1, D))
    if r.integers(3) == 0 and n0 > 2 and n1 > 2:  # duplicates: exact ties
        d0[1] = d0[0]
        d1[2] = d1[0]
    if r.integers(2):
        k = min(n0, n1) // 2
        d1[:k] = d0[:k] + f32(r, (k, D), -0.05, 0.05)  # real correspondences
    d0 /= np.linalg.norm(d0, axis=1, keepdims=True).astype(np.float32)
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True).astype(np.float32)
    d0, d1 = d0.astype(np.float32), d1.astype(np.float32)
    k0, k1 = f32(r, (n0, 3), 0, 250), f32(r, (n1, 3), 0, 250)
    mm = pkg.NearestNeighborMatcher(ratio_thresh=ratio or False, distance_thresh=dist or False, mutual_check=True)
    size = torch.tensor([260, 346])
    f0 = {"sparse_descriptors": t(d0)[None], "sparse_positions": t(k0)[None], "image_size": [size]}
    f1 = {"sparse_descriptors": t(d1)[None], "sparse_positions": t(k1)[None], "image_size": [size]}
    exp = orc.mnn_thresh(d0, d1, ratio, dist) if (ratio or dist) else orc.mnn(d0, d1, want_la=False)
    if not (np.asarray(exp["matches0"]) > -1).any():
        # no match at all: the reference fails in torch.stack([]) (MNN.py:126-127); the drop-in raises the same error
        try:
            mm(f0, f1)
        except RuntimeError as e:
            if "non-empty TensorList" in str(e):
                return
            raise
        raise AssertionError(f"{desc}: no match, but no error")
    got = mm(f0, f1)
    for key in ("matches0", "matches1"):
        if not np.array_equal(n(got[key])[0], exp[key]):
            raise AssertionError(f"{desc}: {key} differs")
    if not np.array_equal(n(got["matching_scores0"])[0], exp["matching_scores0"]):
        raise AssertionError(f"{desc}: matching_scores0 differs")


def sample_case(seed):
    r = np.random.default_rng(seed)
    B, D = int(r.choice([1, 2, 3])), int(r.choice([8, 64, 128, 256]))
    hc, wc = int(r.integers(2, 34)), int(r.integers(2, 45))
    bil = bool(r.integers(2))
    Hp, Wp = (hc * 8, wc * 8) if bil else (hc, wc)
    cap = min(int(r.choice([1, 8, 100])), Hp * Wp)
    counts = r.integers(0, cap + 1, B).astype(np.int32)
    idx = [np.sort(r.choice(Hp * Wp, int(c), replace=False)).astype(np.int32) for c in counts]
    sc = float(r.choice([1.0, 1.41]))
    raw = f32(r, (B, D, hc, wc), -2, 2)
    desc = f"seed {seed}: sample B={B} D={D} {hc}x{wc} bilinear={bil} cap={cap} counts={counts.tolist()}"
    packed = np.zeros((B, cap), np.int32)
    for b in range(B):
        packed[b, :len(idx[b])] = idx[b]
    got = n(N.desc_sample(t(raw), t(packed), t(counts), (Hp, Wp), bilinear=bil, scale=sc))
    exp = orc.desc_sample_bilinear(raw, idx, (Hp, Wp), sc) if bil else orc.desc_gather(raw, idx, sc)
    for b in range(B):
        if not np.array_equal(got[b, :counts[b]], exp[b]):
            raise AssertionError(f"{desc}: image {b} differs")


_LG = {}


def lg_case(seed):
    """LightGlue on ragged small counts: assignments equal except where the oracle's own decision margin is below the float tolerance;
    scores within the north_star's 1e-4."""
    r = np.random.default_rng(seed)
    din = int(r.choice([256, 256, 128]))
    # one case in three: widths other than 256 = 4 x 64 (head_dim = descriptor_dim // num_heads in {32, 64, 128}, lightglue.py:456-461)
    heads, dh, layers = (4, 64, 9) if r.integers(3) else (int(r.choice([1, 2, 3, 4, 6, 8])), int(r.choice([32, 64, 128, 16, 48, 80, 100, 20, 124])), int(r.integers(1, 5)))  # (round 6: any multiple of 4 up to 128)
    key = (din, heads, dh, layers)
    if key not in _LG:
        lg = pkg.LightGlue({"input_dim": din, "descriptor_dim": heads * dh, "num_heads": heads, "n_layers": layers}).to(DEV).eval()
        sd = pkg.synth.synth_state_dict([(k, tuple(v.shape)) for k, v in lg.state_dict().items()], seed=900 + din)
        lg.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        _LG[key] = (lg, sd)
    lg, sd = _LG[key]
    n0, n1 = int(r.choice([1, 2, 3, 31, 64, 65, 130, 257])), int(r.choice([1, 2, 5, 33, 64, 127, 200, 300]))
    H, W = int(r.integers(60, 400)), int(r.integers(60, 500))
    desc = f"seed {seed}: lightglue input_dim={din} heads={heads}x{dh} layers={layers} n0={n0} n1={n1} size {H}x{W}"
    d0, d1 = f32(r, (n0, din)), f32(r, (n1, din))
    k = min(n0, n1) // 2
    d1[:k] = d0[:k] + f32(r, (k, din), -0.1, 0.1)
    d0 = (d0 / np.linalg.norm(d0, axis=1, keepdims=True)).astype(np.float32)
    d1 = (d1 / np.linalg.norm(d1, axis=1, keepdims=True)).astype(np.float32)
    k0 = np.stack([r.uniform(0, H, n0), r.uniform(0, W, n0), r.uniform(0, 1, n0)], 1).astype(np.float32)
    k1 = np.stack([r.uniform(0, H, n1), r.uniform(0, W, n1), r.uniform(0, 1, n1)], 1).astype(np.float32)
    size = torch.tensor([H, W])
    f0 = {"sparse_descriptors": t(d0)[None], "sparse_positions": t(k0)[None], "image_size": [size]}
    f1 = {"sparse_descriptors": t(d1)[None], "sparse_positions": t(k1)[None], "image_size": [size]}
    exp = orc.lightglue(sd, k0, d0, k1, d1, size0=(H, W), size1=(H, W), n_layers=layers, heads=heads)
    if not (np.asarray(exp["matches0"]) > -1).any():
        try:
            lg(f0, f1)
        except RuntimeError as e:
            if "non-empty TensorList" in str(e):
                return
            raise
        raise AssertionError(f"{desc}: no match, but no error")
    got = lg(f0, f1)
    s0 = n(got["matching_scores0"])[0]
    if np.abs(s0 - exp["matching_scores0"]).max() > 1e-4:
        raise AssertionError(f"{desc}: matching_scores0 off by {np.abs(s0 - exp['matching_scores0']).max():.2e}")
    la = n(got["log_assignment"])[0]
    err = np.abs(la - exp["log_assignment"]).max()
    if err > 2e-3:
        raise AssertionError(f"{desc}: log_assignment off by {err:.2e}")
    m0, e0 = n(got["matches0"])[0], np.asarray(exp["matches0"])
    for i in np.nonzero(m0 != e0)[0]:
        # a flipped row must be a near-tie of the oracle's own scores: the two candidates' log-assignments within the tolerance,
        # or the match score within it of the filter threshold
        row = exp["log_assignment"][i, :-1]
        cand = [j for j in (m0[i], e0[i]) if j >= 0]
        near_tie = len(cand) == 2 and abs(row[cand[0]] - row[cand[1]]) < 2e-3
        near_thr = min(abs(float(np.exp(row[j])) - 0.0) for j in cand) < 2e-3 if cand else False
        if not (near_tie or near_thr):
            raise AssertionError(f"{desc}: matches0[{i}] = {m0[i]} vs {e0[i]} without a near-tie")


def lg_batch_case(seed):
    """The batched matcher path (both sides of every pair stacked into one launch per layer, device-side counts): B pairs with
    different keypoint counts each equal their own per-pair oracle run; padding rows beyond a pair's count stay unmatched."""
    from importlib import import_module
    bt = import_module(pkg.__name__ + ".core.modules.matchers._batched")
    r = np.random.default_rng(seed)
    din = 256
    if din not in _LG:  # (the batch case keeps the shipped widths: key = input_dim alone)
        lg = pkg.LightGlue({"input_dim": din}).to(DEV).eval()
        sd = pkg.synth.synth_state_dict([(k, tuple(v.shape)) for k, v in lg.state_dict().items()], seed=900 + din)
        lg.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        _LG[din] = (lg, sd)
    lg, sd = _LG[din]
    B = int(r.choice([2, 3, 5]))
    cap0, cap1 = int(r.choice([8, 64, 150, 256])), int(r.choice([8, 70, 128, 300]))
    n0 = [int(r.integers(1, cap0 + 1)) for _ in range(B)]
    n1 = [int(r.integers(1, cap1 + 1)) for _ in range(B)]
    if r.integers(2):
        n0[0], n1[-1] = cap0, cap1
    H, W = int(r.integers(60, 300)), int(r.integers(60, 400))
    desc = f"seed {seed}: lightglue batch B={B} caps {cap0}/{cap1} counts {n0} / {n1} size {H}x{W}"
    K0, K1 = np.zeros((B, cap0, 3), np.float32), np.zeros((B, cap1, 3), np.float32)
    D0, D1 = np.zeros((B, cap0, din), np.float32), np.zeros((B, cap1, din), np.float32)
    for b in range(B):
        d0, d1 = f32(r, (n0[b], din)), f32(r, (n1[b], din))
        k = min(n0[b], n1[b]) // 2
        d1[:k] = d0[:k] + f32(r, (k, din), -0.1, 0.1)
        D0[b, :n0[b]] = d0 / np.linalg.norm(d0, axis=1, keepdims=True)
        D1[b, :n1[b]] = d1 / np.linalg.norm(d1, axis=1, keepdims=True)
        K0[b, :n0[b]] = np.stack([r.uniform(0, H, n0[b]), r.uniform(0, W, n0[b]), r.uniform(0, 1, n0[b])], 1)
        K1[b, :n1[b]] = np.stack([r.uniform(0, H, n1[b]), r.uniform(0, W, n1[b]), r.uniform(0, 1, n1[b])], 1)
        # rows past the count hold garbage the kernels must never read into a result
        D0[b, n0[b]:] = 7.0
        D1[b, n1[b]:] = -3.0
        K0[b, n0[b]:] = 1e6
    pbs = []
    for K, D, cnt, cap in ((K0, D0, n0, cap0), (K1, D1, n1, cap1)):
        pb = bt.PairBatch()
        pb.kpts, pb.desc, pb.counts = t(K), t(D), t(np.asarray(cnt, np.int32))
        pb.cap, pb.B, pb.image_size, pb.counts_host = cap, B, (H, W), None
        pbs.append(pb)
    mr = lg.match_batched(pbs[0], pbs[1])
    m0, s0 = n(mr.matches0), n(mr.scores0)
    for b in range(B):
        exp = orc.lightglue(sd, K0[b, :n0[b]], D0[b, :n0[b]], K1[b, :n1[b]], D1[b, :n1[b]], size0=(H, W), size1=(H, W))
        if np.abs(s0[b, :n0[b]] - exp["matching_scores0"]).max() > 1e-4:
            raise AssertionError(f"{desc}: pair {b} matching_scores0 off by {np.abs(s0[b, :n0[b]] - exp['matching_scores0']).max():.2e}")
        e0 = np.asarray(exp["matches0"])
        for i in np.nonzero(m0[b, :n0[b]] != e0)[0]:
            row = exp["log_assignment"][i, :-1]
            cand = [j for j in (m0[b, i], e0[i]) if j >= 0]
            near_tie = len(cand) == 2 and abs(row[cand[0]] - row[cand[1]]) < 2e-3
            near_thr = min(abs(float(np.exp(row[j]))) for j in cand) < 2e-3 if cand else False
            if not (near_tie or near_thr):
                raise AssertionError(f"{desc}: pair {b} matches0[{i}] = {m0[b, i]} vs {e0[i]} without a near-tie")
        if (m0[b, n0[b]:] != -1).any() or (m0[b, :n0[b]] >= n1[b]).any():
            raise AssertionError(f"{desc}: pair {b}: a padding row matched or a match points past the other side's count")


def metrics_case(seed):
    from importlib import import_module
    nm = import_module(pkg.__name__ + ".core.metrics._native_metrics")
    r = np.random.default_rng(seed)
    n0, n1 = int(r.choice([0, 1, 5, 200, 1024])), int(r.choice([0, 1, 7, 300, 1024]))
    H0, W0, H1, W1 = int(r.integers(40, 300)), int(r.integers(40, 400)), int(r.integers(40, 300)), int(r.integers(40, 400))
    D = int(r.choice([64, 256]))
    k0 = np.stack([r.uniform(0, H0, n0), r.uniform(0, W0, n0), r.uniform(0, 1, n0)], 1).astype(np.float32)
    hom = None
    if r.integers(2):
        hom = np.eye(3) + r.uniform(-1, 1, (3, 3)) * np.array([[0.1, 0.1, 20], [0.1, 0.1, 20], [2e-4, 2e-4, 0]])
        hom = hom.astype(np.float32)
    k1 = np.stack([r.uniform(0, H1, n1), r.uniform(0, W1, n1), r.uniform(0, 1, n1)], 1).astype(np.float32)
    kk = min(n0, n1) // 2
    if kk:
        k1[:kk, :2] = k0[:kk, :2] + r.uniform(-2, 2, (kk, 2)).astype(np.float32)
    d0, d1 = f32(r, (n0, D)), f32(r, (n1, D))
    nmch = int(r.integers(0, min(n0, n1) + 1)) if min(n0, n1) else 0
    sel = r.choice(min(n0, n1), nmch, replace=False) if nmch else np.zeros((0,), np.int64)
    mk0, mk1 = k0[sel], k1[sel]
    thr = (1, 3) if r.integers(2) else (1, 3, 5)
    desc = f"seed {seed}: metrics n0={n0} n1={n1} matches={nmch} sizes {H0}x{W0} / {H1}x{W1} hom={'yes' if hom is not None else 'no'} thr={thr}"
    exp = orc.pair_metrics(k0, k1, d0, d1, mk0, mk1, (H0, W0), (H1, W1), None if hom is None else hom.reshape(-1).tolist(), mma_thr=thr, vdd_thr=thr)
    gd = nm.single_pair(t(k0), t(k1), t(d0), t(d1), t(mk0), t(mk1), (H0, W0), (H1, W1), None if hom is None else t(hom), thr, thr)
    got = np.array([gd[k] for k in nm.metric_names(thr, thr)])
    if got.shape != exp.shape or not np.allclose(got, exp, atol=2e-3, rtol=1e-5, equal_nan=True):
        raise AssertionError(f"{desc}: {got} vs {exp}")


CASES = [conv_case, conv_case, detect_case, upsample_case, mnn_case, sample_case, lg_case, metrics_case, lg_batch_case]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed0", type=int, default=1)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    cases = [c for c in CASES if a.only in c.__name__]
    t0, seed, ok, bad, last = time.time(), a.seed0, 0, [], time.time()
    per = {}
    while time.time() - t0 < a.seconds:
        fn = cases[seed % len(cases)]
        try:
            fn(seed)
            ok += 1
            per[fn.__name__] = per.get(fn.__name__, 0) + 1
        except AssertionError as e:
            bad.append(str(e))
            print("MISMATCH", e, flush=True)
        except Exception as e:
            bad.append(f"seed {seed} ({fn.__name__}): {type(e).__name__}: {str(e)[:300]}")
            print("ERROR seed", seed, fn.__name__, type(e).__name__, str(e)[:300], flush=True)
            traceback.print_exc(limit=4)
        seed += 1
        if time.time() - last > 45:
            last = time.time()
            print(f"... {ok} cases equal, {len(bad)} findings, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz_ops: {ok} cases bit-equal to the oracle {per}, {len(bad)} findings, seeds {a.seed0}..{seed - 1}")
    for b in bad[:40]:
        print("  ", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
