"""Plain-PyTorch CPU expression of the SP+MNN pipeline (the second CPU timing beside the C oracle).

TEST / MEASUREMENT INFRASTRUCTURE ONLY (like everything under oracle/): bench.py --cpu-torch times it on the
host cores and tests/test_oracle_golden.py checks it loosely against the C oracle.  It is written from the
stage descriptions in SURVEY.md section 8a, with torch's own operators (MKL-DNN convolutions, max_pool2d,
softmax, pixel_shuffle, sort, grid_sample, matmul), i.e. what the reference's CPU path executes, without any
of the reference's files."""
import torch
import torch.nn.functional as F


def _blocks(sd, prefix, names, bn):
    """[(weight, bias, bn-tuple or None)] for the conv blocks called `names` under `prefix`"""
    out = []
    for n in names:
        w, b = sd[f"{prefix}{n}.0.weight" if bn else f"{prefix}{n}.weight"], sd[f"{prefix}{n}.0.bias" if bn else f"{prefix}{n}.bias"]
        stats = None
        if bn:
            k = f"{prefix}{n}.2." if f"{prefix}{n}.2.weight" in sd else f"{prefix}{n}.1."
            stats = tuple(sd[k + s] for s in ("weight", "bias", "running_mean", "running_var"))
        out.append((w, b, stats))
    return out


def _conv(x, w, b, stats, relu=True):
    x = F.conv2d(x, w, b, padding=w.shape[-1] // 2)
    if relu:
        x = F.relu(x)
    if stats is not None:
        g, be, mean, var = stats
        x = F.batch_norm(x, mean, var, g, be, training=False, eps=1e-5)
    return x


def vgg_event_net(sd, x):
    """VGGExtractor network (event side): 8 x [conv3x3 -> ReLU -> BN] with pools after blocks 2, 4, 6; two heads."""
    p = "backbone."
    names = ["l1.0", "l1.1", "l2.0", "l2.1", "l3.0", "l3.1", "l4.0", "l4.1"]
    for i, (w, b, st) in enumerate(_blocks(sd, p, names, True)):
        x = _conv(x, w, b, st)
        if i in (1, 3, 5):
            x = F.max_pool2d(x, 2, 2)
    feats = x
    (w1, b1, s1), = _blocks(sd, "detector_head.", ["_detH1"], True)
    d = _conv(feats, w1, b1, s1)
    d = F.batch_norm(F.conv2d(d, sd["detector_head._detH2.0.weight"], sd["detector_head._detH2.0.bias"]),
                     sd["detector_head._detH2.1.running_mean"], sd["detector_head._detH2.1.running_var"],
                     sd["detector_head._detH2.1.weight"], sd["detector_head._detH2.1.bias"], training=False, eps=1e-5)
    (w2, b2, s2), = _blocks(sd, "descriptor_head.", ["_desH1"], True)
    r = _conv(feats, w2, b2, s2)
    r = F.batch_norm(F.conv2d(r, sd["descriptor_head._desH2.0.weight"], sd["descriptor_head._desH2.0.bias"]),
                     sd["descriptor_head._desH2.1.running_mean"], sd["descriptor_head._desH2.1.running_var"],
                     sd["descriptor_head._desH2.1.weight"], sd["descriptor_head._desH2.1.bias"], training=False, eps=1e-5)
    return d, r


def superpoint_net(sd, x):
    for i, n in enumerate(["conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b"]):
        x = F.relu(F.conv2d(x, sd[n + ".weight"], sd[n + ".bias"], padding=1))
        if i in (1, 3, 5):
            x = F.max_pool2d(x, 2, 2)
    d = F.conv2d(F.relu(F.conv2d(x, sd["convPa.weight"], sd["convPa.bias"], padding=1)), sd["convPb.weight"], sd["convPb.bias"])
    r = F.conv2d(F.relu(F.conv2d(x, sd["convDa.weight"], sd["convDa.bias"], padding=1)), sd["convDb.weight"], sd["convDb.bias"])
    return d, r


def nms_fixpoint(score, radius=4):
    """first-maximum-wins (2r+1)^2 suppression iterated to its fix-point"""
    k = 2 * radius + 1
    B, _, H, W = score.shape
    count = -1
    while True:
        patches = F.unfold(score, k, padding=radius)  # [B, k*k, H*W]
        is_max = (patches.argmax(1) == (k * k) // 2) & (score.reshape(B, -1) > 0)
        n = int(is_max.sum())
        if n == count:
            return score
        count = n
        m = is_max.reshape(B, 1, H, W).float()
        near = F.max_pool2d(m, k, 1, radius) > 0
        score = torch.where(near & ~is_max.reshape(B, 1, H, W), torch.zeros_like(score), score)


def extract(kind, sd, x, mask, top_k=1024, border=4, scale=1.0, dense=False):
    """-> per-image (positions [n,3] yx+score, descriptors [n,256]); dense=True also builds the reference's dense outputs
    (normalized_descriptors: the raw map resized to the padded image size, L2-normalised, x scale -- what dominates the
    reference's CPU time)"""
    B, _, H, W = x.shape
    ph, pw = (-H) % 8, (-W) % 8
    h0, w0 = ph // 2, pw // 2
    if kind == "superpointv1":
        x = x / 255.0
    xp = F.pad(x, (w0, pw - w0, h0, ph - h0), mode="replicate")
    logits, raw = (vgg_event_net if kind == "vgg" else superpoint_net)(sd, xp)
    prob = F.softmax(logits, 1)
    score = F.pixel_shuffle(prob[:, :64], 8)
    Hp, Wp = score.shape[-2:]
    if mask is not None:
        m = F.pad(mask.float(), (w0, pw - w0, h0, ph - h0))
        score = score * (F.max_pool2d(m, 3, 1, 1) > 0)
    score[:, :, :border] = 0
    score[:, :, -border:] = 0
    score[:, :, :, :border] = 0
    score[:, :, :, -border:] = 0
    if dense:
        up = F.normalize(F.interpolate(raw, size=score.shape[-2:], mode="bilinear", align_corners=False), dim=1) * scale
        up = up[:, :, h0:h0 + H, w0:w0 + W].contiguous()  # un-padded
        del up
    nms = nms_fixpoint(score)
    flat = nms.reshape(B, -1)
    N = flat.shape[1]
    srt = flat.sort(1).values
    thr = (srt[:, N - top_k - 1] + srt[:, N - top_k]) * 0.5 if top_k < N else torch.zeros(B)
    out = []
    for b in range(B):
        idx = torch.nonzero(flat[b] > thr[b]).squeeze(1)
        y, xx = (idx // Wp).float(), (idx % Wp).float()
        grid = torch.stack([2 * (xx / (Wp - 1)) - 1, 2 * (y / (Hp - 1)) - 1], -1)[None, None]
        d = F.grid_sample(raw[b:b + 1], grid, mode="bilinear", align_corners=False)[0, :, 0].t()
        d = F.normalize(d, dim=1) * scale
        pos = torch.stack([y + 0.5 - h0, xx + 0.5 - w0, flat[b][idx]], 1)
        out.append((pos, d))
    return out


def mnn(d0, d1):
    sim = d0 @ d1.t()
    m0, m1 = sim.argmax(1), sim.argmax(0)
    ar = torch.arange(len(d0))
    return torch.where(m1[m0] == ar, m0, torch.full_like(m0, -1))


@torch.no_grad()
def sp_mnn_pairs(sd_event, sd_image, events, mask, image, top_k=1024, dense=False):
    """the whole SP+MNN pipeline for a batch; returns the number of mutual matches per pair"""
    te = {k: torch.from_numpy(v) for k, v in sd_event.items()}
    ti = {k: torch.from_numpy(v) for k, v in sd_image.items()}
    fe = extract("vgg", te, torch.from_numpy(events), torch.from_numpy(mask), top_k, dense=dense)
    fi = extract("superpointv1", ti, torch.from_numpy(image), None, top_k, dense=dense)
    return [int((mnn(a[1], b[1]) > -1).sum()) for a, b in zip(fe, fi)], fe, fi
