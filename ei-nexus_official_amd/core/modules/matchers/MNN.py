"""Mutual-nearest-neighbour matcher, native on MI355X (csrc/mnn.hip).

Drop-in for NearestNeighborMatcher (reference core/modules/matchers/MNN.py:35-140): same
constructor and the same output dict, minus `similarity` unless `return_similarity` is set (the
reference's Matcher wrapper drops it anyway, Matchers.py:177-186).  The ratio / distance
thresholds of find_nn (MNN.py:12-22; off in every shipped config, configs/model/SP_MNN.yaml:65-66)
run as a second-neighbour tile pass + a masked finalize (einx_mnn_thresh).
"""
import torch
from torch import nn

from ...._native import on_input_device
from .... import _native as N
from ._batched import from_feats, materialize_matches, stacked_outputs


class NearestNeighborMatcher(nn.Module):
    def __init__(self, ratio_thresh=None, distance_thresh=None, mutual_check=True):
        super().__init__()
        if not mutual_check:
            raise NotImplementedError("einx MNN always applies the mutual check, as every EI-Nexus config does")
        self.ratio_thresh = ratio_thresh
        self.distance_thresh = distance_thresh
        self.mutual_check = mutual_check
        self.want_log_assignment = True
        self.return_similarity = False

    @on_input_device
    def match_batched(self, pb0, pb1):
        """device-side: no host sync"""
        return N.mnn(pb0.desc, pb0.counts, pb1.desc, pb1.counts, want_la=self.want_log_assignment, ratio_thresh=self.ratio_thresh,
                     distance_thresh=self.distance_thresh, gather=(pb0.kpts, pb1.kpts, 3))

    @torch.no_grad()
    @on_input_device
    def forward(self, feats0, feats1):
        """B == 1: the per-pair dict of the frozen path.  B > 1 (stacked [B,n,*] inputs, every pair
        with the same n and m -- the un-frozen Matcher branch): whole-batch tensors, lists of matched
        keypoints and `similarity`, as MNN.py:103-140 returns them."""
        pos0, pos1 = feats0["sparse_positions"], feats1["sparse_positions"]
        stacked = torch.is_tensor(pos0) and pos0.dim() == 3 and torch.is_tensor(pos1) and pos1.dim() == 3
        if stacked and pos0.shape[0] > 1 and (pos0.numel() == 0 or pos1.numel() == 0):
            f = feats0["sparse_descriptors"]
            B, n, m = pos0.shape[0], pos0.shape[1], pos1.shape[1]
            print("No keypoints found in either image")
            return {"matches0": f.new_full((B, n), -1), "matches1": f.new_full((B, m), -1), "matching_scores0": f.new_zeros((B, n)),
                    "matching_scores1": f.new_zeros((B, m)), "matched_kpts0": [f.new_zeros((n, 3))] * B,
                    "matched_kpts1": [f.new_zeros((m, 3))] * B, "similarity": f.new_zeros((B, n, m)),
                    "log_assignment": f.new_zeros((B, n + 1, m + 1))}
        pb0, pb1 = from_feats(feats0), from_feats(feats1)
        r = self.match_batched(pb0, pb1)
        nm = r.nmatch.cpu().tolist()
        n = pb0.counts_host or pb0.counts.cpu().tolist()
        m = pb1.counts_host or pb1.counts.cpu().tolist()
        if self.ratio_thresh and any((0 < a < 2 or 0 < b < 2) and a > 0 and b > 0 for a, b in zip(n, m)):
            raise RuntimeError("selected index k out of range")  # sim.topk(2) on a single candidate (MNN.py:13-14)
        if pb0.B == 1:
            out = {k: v[0] for k, v in materialize_matches(r, n, m, nm, 3).items()}
            if self.return_similarity and n[0] and m[0]:
                out["similarity"] = N.similarity(pb0.desc, pb0.counts, pb1.desc, pb1.counts)[:, :n[0], :m[0]]
            return out
        if not (stacked and len(set(n)) == 1 and len(set(m)) == 1):
            raise NotImplementedError("einx MNN.forward with B > 1 takes stacked [B,n,*] tensors (as the reference does); "
                                      "use Matcher for ragged batches")
        out = stacked_outputs(r, nm, 3)
        out["similarity"] = N.similarity(pb0.desc, pb0.counts, pb1.desc, pb1.counts)
        if out["log_assignment"] is None:
            del out["log_assignment"]
        return out
