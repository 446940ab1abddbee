"""Mutual-nearest-neighbour matcher, native on MI355X (csrc/mnn.hip).

Drop-in for NearestNeighborMatcher (reference core/modules/matchers/MNN.py:35-140): same
constructor and the same output dict, minus `similarity` unless `return_similarity` is set (the
reference's Matcher wrapper drops it anyway, Matchers.py:177-186).  The ratio / distance
thresholds are disabled in every shipped config (configs/model/SP_MNN.yaml:65-66) and are not
implemented.
"""
import torch
from torch import nn

from .... import _native as N
from ._batched import from_feats, materialize_matches


class NearestNeighborMatcher(nn.Module):
    def __init__(self, ratio_thresh=None, distance_thresh=None, mutual_check=True):
        super().__init__()
        if ratio_thresh or distance_thresh:
            raise NotImplementedError("einx MNN implements the shipped configuration (no ratio / distance threshold)")
        if not mutual_check:
            raise NotImplementedError("einx MNN always applies the mutual check, as every EI-Nexus config does")
        self.ratio_thresh = ratio_thresh
        self.distance_thresh = distance_thresh
        self.mutual_check = mutual_check
        self.want_log_assignment = True

    def match_batched(self, pb0, pb1):
        """device-side: no host sync"""
        r = N.mnn(pb0.desc, pb0.counts, pb1.desc, pb1.counts, want_la=self.want_log_assignment)
        return N.gather_matches(r, pb0.kpts, pb1.kpts, pb0.counts, 3)

    @torch.no_grad()
    def forward(self, feats0, feats1):
        pb0, pb1 = from_feats(feats0), from_feats(feats1)
        r = self.match_batched(pb0, pb1)
        nm = r.nmatch.cpu().tolist()
        n = pb0.counts_host or pb0.counts.cpu().tolist()
        m = pb1.counts_host or pb1.counts.cpu().tolist()
        lists = materialize_matches(r, n, m, nm, 3)
        if pb0.B == 1:
            return {k: v[0] for k, v in lists.items()}
        out = {k: (v if k.startswith("matched") else torch.cat(v, 0) if len(set(t.shape for t in v)) == 1 else v) for k, v in lists.items()}
        return out
