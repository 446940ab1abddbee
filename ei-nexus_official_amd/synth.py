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


def synth_unit_descriptors(seed, n, dim, scale=1.0):
    d = normalish(seed, (n, dim), lane=5).astype(np.float64)
    d = d / np.maximum(np.sqrt((d * d).sum(-1, keepdims=True)), 1e-12)
    return (d * scale).astype(np.float32)
