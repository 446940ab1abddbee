"""Device-side batching helpers shared by the native matchers."""
import torch

from ...._extract import FeatsDict


class PairBatch:
    """Keypoints/descriptors of B images packed as [B,cap,*] with device-side counts."""
    __slots__ = ("kpts", "desc", "counts", "cap", "B", "image_size", "counts_host")


def from_batched(bf):
    pb = PairBatch()
    pb.kpts, pb.desc, pb.counts = bf.det.positions, bf.sparse_desc, bf.det.counts
    pb.cap, pb.B, pb.image_size = bf.det.cap, bf.B, bf.image_size
    pb.counts_host = None
    return pb


def _lists_untouched(feats, pos, desc):
    """The hidden device batch may stand in for the lists only while the lists still ARE views of it: a caller
    may legally filter / replace feats['sparse_positions'] / ['sparse_descriptors'] between the extractor and
    the matcher (the reference reads the lists), and then the lists win."""
    bf = feats._batched
    if torch.is_tensor(pos) or torch.is_tensor(desc) or len(pos) != bf.B or len(desc) != bf.B:
        return False
    kp, ds, ns = bf.det.positions, bf.sparse_desc, getattr(bf, "_ns", None)
    if ns is None:
        return False
    for b in range(bf.B):
        p, d = pos[b], desc[b]
        if p.shape[0] != ns[b] or d.shape[0] != ns[b] or p.data_ptr() != kp[b].data_ptr() or d.data_ptr() != ds[b].data_ptr():
            return False
        if p.shape[0] and (p.stride(0) != kp.stride(1) or d.stride(0) != ds.stride(1)):
            return False
    return True


def from_feats(feats):
    """Accepts the reference-style dict (lists of per-image tensors, or stacked tensors)."""
    pos, desc = feats["sparse_positions"], feats["sparse_descriptors"]
    if torch.is_tensor(pos) and pos.dim() == 3 and pos.shape[1] > 0:
        # stacked [B,n,*] tensors (the un-frozen Matcher branch, Matchers.py:145-149): used as they are
        B, n = int(pos.shape[0]), int(pos.shape[1])
        pb = PairBatch()
        if pos.shape[2] == 3:
            pb.kpts = pos.to(torch.float32).contiguous()
        else:
            pb.kpts = torch.zeros((B, n, 3), dtype=torch.float32, device=pos.device)
            pb.kpts[:, :, :pos.shape[2]] = pos
        pb.desc = desc.to(torch.float32).contiguous()
        pb.counts_host = [n] * B
        pb.counts = torch.full((B,), n, dtype=torch.int32, device=pos.device)
        size = feats["image_size"][0]
        pb.image_size = (int(size[0]), int(size[1]))
        pb.cap, pb.B = n, B
        return pb
    if isinstance(feats, FeatsDict) and feats._batched is not None and _lists_untouched(feats, pos, desc):
        pb = from_batched(feats._batched)
        pb.counts_host = [int(p.shape[0]) for p in feats["sparse_positions"]]
        return pb
    if torch.is_tensor(pos):
        pos, desc = list(pos), list(desc)
    B = len(pos)
    cap = max([int(p.shape[0]) for p in pos] + [1])
    dev, D = desc[0].device, int(desc[0].shape[-1])
    pb = PairBatch()
    pb.kpts = torch.zeros((B, cap, 3), dtype=torch.float32, device=dev)
    pb.desc = torch.zeros((B, cap, D), dtype=torch.float32, device=dev)
    pb.counts_host = [int(p.shape[0]) for p in pos]
    for b in range(B):
        n = pb.counts_host[b]
        if n:
            pb.kpts[b, :n, :pos[b].shape[1]] = pos[b]
            pb.desc[b, :n] = desc[b]
    pb.counts = torch.tensor(pb.counts_host, dtype=torch.int32, device=dev)
    size = feats["image_size"][0]
    pb.image_size = (int(size[0]), int(size[1]))
    pb.cap, pb.B = cap, B
    return pb


def full_batch_lists(r):
    """count-independent part of the per-pair lists when every image filled its quota (one unbind per
    output); can be built before the host knows the counts."""
    B = r.matches0.shape[0]
    return {
        "matches0": list(r.matches0[:, None, :].unbind(0)), "matches1": list(r.matches1[:, None, :].unbind(0)),
        "matching_scores0": list(r.scores0[:, None, :].unbind(0)), "matching_scores1": list(r.scores1[:, None, :].unbind(0)),
        "log_assignment": [None] * B if r.la is None else list(r.la[:, None].unbind(0)),
    }


def materialize_matches(r, n_host, m_host, nmatch_host, cols, extra=None, prebuilt=None):
    """MatchResult (device, padded) -> the reference's per-pair lists (Matchers.py:168-203).
    Reproduces the empty-input dict (MNN.py:63-86 / lightglue.py:569-591) and the zero-match
    `torch.stack([])` failure (MNN.py:126-127) of the reference."""
    keys = ("matches0", "matches1", "matching_scores0", "matching_scores1", "matched_kpts0", "matched_kpts1", "log_assignment")
    out = {k: [] for k in keys}
    if extra:
        for k in extra:
            out[k] = []
    B = len(n_host)
    cap0, cap1 = r.matches0.shape[1], r.matches1.shape[1]
    if all(v == cap0 for v in n_host) and all(v == cap1 for v in m_host) and all(v > 0 for v in nmatch_host) and cap0 > 0 and cap1 > 0:
        # common case: every image filled its keypoint quota -> whole-batch views, per-pair work only
        # where the length really differs (matched keypoints)
        pre = prebuilt if prebuilt is not None else full_batch_lists(r)
        out.update(pre)
        if r.mk0_flat is not None:  # packed on the device: one split instead of 2 B slicing calls
            total = sum(nmatch_host)
            out["matched_kpts0"] = list(r.mk0_flat[:total].split(nmatch_host))
            out["matched_kpts1"] = list(r.mk1_flat[:total].split(nmatch_host))
        else:
            out["matched_kpts0"] = [r.mk0[b, :nmatch_host[b], :cols] for b in range(B)]
            out["matched_kpts1"] = [r.mk1[b, :nmatch_host[b], :cols] for b in range(B)]
        return out
    for b in range(B):
        n, m = n_host[b], m_host[b]
        if n == 0 or m == 0:
            print("No keypoints found in either image")
            f = r.scores0
            out["matches0"].append(f.new_full((1, n), -1))
            out["matches1"].append(f.new_full((1, m), -1))
            out["matching_scores0"].append(f.new_zeros((1, n)))
            out["matching_scores1"].append(f.new_zeros((1, m)))
            out["matched_kpts0"].append(f.new_zeros((0, 3)))
            out["matched_kpts1"].append(f.new_zeros((0, 3)))
            out["log_assignment"].append(f.new_zeros((1, n + 1, m + 1)))
            continue
        M = nmatch_host[b]
        if M == 0:
            raise RuntimeError("stack expects a non-empty TensorList")  # what torch.stack([]) raises in the reference
        out["matches0"].append(r.matches0[b, :n][None])
        out["matches1"].append(r.matches1[b, :m][None])
        out["matching_scores0"].append(r.scores0[b, :n][None])
        out["matching_scores1"].append(r.scores1[b, :m][None])
        out["matched_kpts0"].append(r.mk0[b, :M, :cols])
        out["matched_kpts1"].append(r.mk1[b, :M, :cols])
        out["log_assignment"].append(None if r.la is None else r.la[b, :n + 1, :m + 1][None])
    return out


def stacked_outputs(r, nmatch_host, cols, mk0=None, mk1=None):
    """The reference matchers' own return value for a stacked batch with b > 1
    (MNN.py:103-118,131-140; lightglue.py:675-687,700-712): whole-batch tensors plus per-pair lists
    of matched keypoints.  A pair without any match makes the reference's torch.stack([]) raise."""
    if any(v == 0 for v in nmatch_host):
        raise RuntimeError("stack expects a non-empty TensorList")
    mk0 = r.mk0 if mk0 is None else mk0
    mk1 = r.mk1 if mk1 is None else mk1
    B = len(nmatch_host)
    return {
        "matches0": r.matches0, "matches1": r.matches1, "matching_scores0": r.scores0, "matching_scores1": r.scores1,
        "matched_kpts0": [mk0[b, :nmatch_host[b], :cols] for b in range(B)],
        "matched_kpts1": [mk1[b, :nmatch_host[b], :cols] for b in range(B)],
        "log_assignment": r.la,
    }


