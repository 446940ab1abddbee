"""Deterministic synthetic tensors (weights, event voxels, images, descriptors).

Everything is derived from integer hashing (splitmix64) done in numpy uint64
arithmetic followed by exactly-representable fp32 scaling, so the same
(seed, shape) gives the same bits on every machine and numpy version -- the
golden-fixture generator (tests/golden/gen_golden.py), the parity tests and
bench.py all draw from here.  No libm calls (no log/cos), hence no platform
drift.

Input shapes follow SURVEY.md section 8d: events [B,Ce,260,346] sparse voxel
grid with ~10 % support shared across bins, events_mask = support, image
[B,1,260,346] 3x3-box-filtered integers in 0..255.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _stream(seed, n, lane=0):
    base = np.uint64((int(seed) * 0x2545F4914F6CDD1D + int(lane) * 0xD1342543DE82EF95) & 0xFFFFFFFFFFFFFFFF)
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _splitmix64(_splitmix64(idx + base))


def uniform01(seed, shape, lane=0):
    """fp32 uniform on [0,1) with 24 random bits (exact in fp32)."""
    n = int(np.prod(shape)) if len(shape) else 1
    h = _stream(seed, n, lane)
    u = (h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return u.reshape(shape)


def uniform(seed, shape, lo, hi, lane=0):
    u = uniform01(seed, shape, lane)
    return (np.float32(lo) + u * np.float32(hi - lo)).astype(np.float32)


def normalish(seed, shape, lane=0):
    """Approximately N(0,1): centred Irwin-Hall sum of 4 uniforms, exact fp32 arithmetic."""
    acc = np.zeros(shape, np.float32)
    for j in range(4):
        acc = acc + uniform01(seed, shape, lane * 4 + j + 101)
    return ((acc - np.float32(2.0)) * np.float32(1.7320508)).astype(np.float32)


def name_seed(name, seed):
    return (zlib.crc32(name.encode()) ^ (int(seed) * 0x9E3779B1)) & 0x7FFFFFFF


def synth_param(name, shape, seed=0):
    """One state-dict entry from its key name and shape (see gen_golden.py for use)."""
    s = name_seed(name, seed)
    shape = tuple(int(v) for v in shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, np.int64)
    if leaf == "running_mean":
        return uniform(s, shape, -0.3, 0.3)
    if leaf == "running_var":
        return uniform(s, shape, 0.5, 1.5)
    if name.endswith("posenc.Wr.weight"):
        return normalish(s, shape)
    if leaf == "weight":
        if len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            a = float(np.sqrt(6.0 / fan_in))
            return uniform(s, shape, -a, a)
        if len(shape) == 2:
            a = float(np.sqrt(3.0 / shape[1]))
            return uniform(s, shape, -a, a)
        return uniform(s, shape, 0.5, 1.5)  # BatchNorm / LayerNorm gain
    if leaf == "bias":
        return uniform(s, shape, -0.1, 0.1)
    return uniform(s, shape, -0.1, 0.1)


def synth_state_dict(keys_shapes, seed=0, skip=("descriptor_scale_factor",)):
    """keys_shapes: iterable of (key, shape).  Returns {key: np.ndarray}."""
    out = {}
    for k, shp in keys_shapes:
        if any(k.endswith(sfx) for sfx in skip):
            continue
        out[k] = synth_param(k, shp, seed)
    return out


def synth_events(seed, batch, channels, height=260, width=346, support=0.10):
    """events [B,C,H,W] fp32 (zeros off-support, ~N(0,1) on it) and events_mask [B,1,H,W] bool."""
    ev = np.zeros((batch, channels, height, width), np.float32)
    mask = np.zeros((batch, 1, height, width), bool)
    for b in range(batch):
        s = 1234 + seed + b
        sup = uniform01(s, (height, width), lane=1) < np.float32(support)
        val = normalish(s, (channels, height, width), lane=2)
        ev[b] = np.where(sup[None], val, np.float32(0.0))
        mask[b, 0] = sup
    return ev, mask


def synth_image(seed, batch, height=260, width=346):
    """image [B,1,H,W] fp32 in 0..255: 3x3 box filter (edge-replicated) of uniform integers."""
    img = np.zeros((batch, 1, height, width), np.float32)
    for b in range(batch):
        s = 1234 + seed + b
        raw = np.floor(uniform01(s, (height, width), lane=3) * np.float32(256.0)).astype(np.float32)
        p = np.pad(raw, 1, mode="edge")
        acc = np.zeros((height, width), np.float32)
        for dy in range(3):
            for dx in range(3):
                acc = acc + p[dy:dy + height, dx:dx + width]
        img[b, 0] = acc / np.float32(9.0)
    return img


def synth_raw_events(seed, n, height=260, width=346, hot=0.3):
    """raw sensor events {"x","y","t","p"} (integer pixel coordinates, increasing float64 timestamps, polarity +-1) -- what the
    reference's dataset hands datasets/representations.py::events_to_voxel_grid.  A share `hot` of them clusters in the
    middle eighth of the sensor so that the accumulation image has a wide count range."""
    u = uniform01(seed, (n,)).astype(np.float64)
    t = 1.5e9 + np.cumsum(u * 1e-4 + 1e-6)
    x = np.floor(uniform01(seed + 1, (n,)) * np.float32(width - 1))
    y = np.floor(uniform01(seed + 2, (n,)) * np.float32(height - 1))
    h = uniform01(seed + 4, (n,)) < np.float32(hot)
    x = np.where(h, np.float32(width // 2) + np.floor(x / 8), x).astype(np.float32)
    y = np.where(h, np.float32(height // 2) + np.floor(y / 8), y).astype(np.float32)
    p = np.where(uniform01(seed + 3, (n,)) < np.float32(0.5), np.float32(1.0), np.float32(-1.0)).astype(np.float32)
    return {"x": x, "y": y, "t": t, "p": p}


def synth_unit_descriptors(seed, n, dim, scale=1.0):
    d = normalish(seed, (n, dim), lane=5).astype(np.float64)
    d = d / np.maximum(np.sqrt((d * d).sum(-1, keepdims=True)), 1e-12)
    return (d * scale).astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------------
# "Same scene" workloads (round 4).  Independent random networks on independent random inputs give descriptors that are
# nearly orthogonal across the two sides (a handful of mutual nearest neighbours in 1024, LightGlue scores ~1e-6), a
# regime in which float tolerances on the matcher outputs check nothing.  The recipes below are RULES on top of the
# name-synthesised weights (no stored data): the event extractor computes the image extractor's function (same conv
# weights, identity BatchNorm where the image side has none), and the event voxels are the image / 255 in every bin
# plus a sparse perturbation -- events and image of one scene.  Hundreds of mutual matches per pair follow.
# ---------------------------------------------------------------------------------------------------------------------
_EV = "event_extractor.extractor."
_IM = "image_extractor.extractor."
_EV_SLOTS = [f"backbone.l{s}.{i}" for s in (1, 2, 3, 4) for i in (0, 1)] + \
    ["detector_head._detH1", "detector_head._detH2", "descriptor_head._desH1", "descriptor_head._desH2"]
_SP_SLOTS = [f"conv{s}{ab}" for s in (1, 2, 3, 4) for ab in "ab"] + ["convPa", "convPb", "convDa", "convDb"]
_SILK_HEADS = "model.backbone._heads._mods."
_SILK_SLOTS = [f"model.backbone._backbone.layers.{s}.{i}" for s in (0, 1, 2, 3) for i in (0, 1)] + \
    [_SILK_HEADS + "logits._detH1", _SILK_HEADS + "logits._detH2", _SILK_HEADS + "raw_descriptors._desH1",
     _SILK_HEADS + "raw_descriptors._desH2"]


def _embed(dst_shape, src, fill=0.0):
    out = np.full(dst_shape, np.float32(fill), np.float32)
    out[tuple(slice(0, s) for s in src.shape)] = src
    return out


def twin_overrides(sd):
    """sd: numpy state dict of an EIM (event VGG / VGG_NP + image SuperPointv1 / SiLK).  Returns the event-extractor entries
    that make it the image extractor's twin: conv i <- image conv i (a thinner image layer is embedded in the leading
    channels, the rest zero; a 1-channel first layer is spread evenly over the event bins), BatchNorm <- the image side's
    BatchNorm (SiLK) or the identity (SuperPoint has none)."""
    silk = any(k.startswith(_IM + "model.backbone.") for k in sd)
    out = {}
    for es, ims in zip(_EV_SLOTS, _SILK_SLOTS if silk else _SP_SLOTS):
        ew = _EV + es + ".0.weight"
        iw = _IM + ims + (".0.weight" if silk else ".weight")
        w = sd[iw]
        shape = sd[ew].shape
        if w.shape[1] == 1 and shape[1] > 1:
            w = (np.repeat(w, shape[1], 1) / np.float32(shape[1])).astype(np.float32)
        out[ew] = _embed(shape, w)
        out[_EV + es + ".0.bias"] = _embed(shape[:1], sd[iw[:-6] + "bias"])
        ebn = _EV + es + (".2." if _EV + es + ".2.weight" in sd else ".1.")
        ibn = _IM + ims + (".2." if _IM + ims + ".2.weight" in sd else ".1.")
        for leaf, ident in (("weight", 1.0), ("bias", 0.0), ("running_mean", 0.0), ("running_var", 1.0)):
            out[ebn + leaf] = _embed(shape[:1], sd[ibn + leaf], ident) if silk else np.full(shape[:1], np.float32(ident), np.float32)
    return out


def twin_events(ev, img, alpha=0.05):
    """event voxels of the same scene: image / 255 in every bin + alpha x the sparse synthetic events"""
    return (ev * np.float32(alpha) + img / np.float32(255.0)).astype(np.float32)


def lightglue_calibration(sd, x, prefix="", temperature=32.0, z_mean=3.0, layer=8):
    """Random-weight LightGlue ends in descriptors dominated by one common vector (norm ~47 of ~48), so its assignment is
    flat.  Remove it where a trained network would not have it in the first place -- in the last block's residual update
    (cross_attn.ffn.3.bias <- bias - mean(x), the same vector on both sides: lightglue.py:327-328) --, scale final_proj so that a
    typical centred descriptor has |mdesc|^2 = `temperature` (lightglue.py:391-394: sim = <W x0 + b, W x1 + b> / 16) and shift
    the matchability logit to a mean of `z_mean`.  x: [N,256] final-layer descriptors of both sides (the matcher's
    ref_descriptors).  Returns (overrides, scale): final_proj.weight is W * float32(scale) elementwise (a rule; fixtures store
    the scalar), final_proj.bias is scaled alike, the two bias vectors are data."""
    p = f"{prefix}log_assignment.{layer}."
    fb = f"{prefix}transformers.{layer}.cross_attn.ffn.3.bias"
    x = np.asarray(x, np.float64)
    mean = x.mean(0)
    xc = x - mean
    cn = float(np.sqrt((xc ** 2).sum(1)).mean())
    scale = np.float32(np.sqrt(temperature) * 4.0 / cn)
    z = xc @ sd[p + "matchability.weight"].astype(np.float64).reshape(-1)
    return {fb: (sd[fb].astype(np.float64) - mean).astype(np.float32),
            p + "final_proj.weight": (sd[p + "final_proj.weight"] * scale).astype(np.float32),
            p + "final_proj.bias": (sd[p + "final_proj.bias"] * scale).astype(np.float32),
            p + "matchability.bias": np.array([z_mean - float(z.mean())], np.float32)}, scale
