"""Matcher wrapper (reference core/modules/Matchers.py:13-222): config dispatch and the frozen
inference branch.  The reference calls its matcher once per sample; here the whole batch goes to
the device in one launch sequence and the per-sample lists are cut afterwards.  The non-frozen
branch (random padding to max_points_num, Matchers.py:204-222) is training-only, RNG dependent and
out of scope."""
import torch
from torch import nn

from .matchers.MNN import NearestNeighborMatcher
from .matchers.lightglue import LightGlue
from .matchers._batched import from_batched, from_feats, materialize_matches


class Matcher(nn.Module):
    def __init__(self, config, logger=None, device="cuda"):
        super().__init__()
        self.config = config.matcher
        self.matcher = None
        self.freeze = self.config.freeze
        self.max_points_num = self.config.max_points_num
        self.pad_mode = self.config.pad_mode
        self.desc_scale_factor = self.config.desc_scale_factor
        self.matcher_type = self.config.type
        if self.matcher_type == "MNN":
            self.matcher = NearestNeighborMatcher(ratio_thresh=self.config.MNN.ratio_thresh,
                                                  distance_thresh=self.config.MNN.distance_thresh, mutual_check=True)
        elif self.matcher_type == "LightGlue":
            self.matcher = LightGlue(conf=self.config.LightGlue)
        elif self.matcher_type is None:
            self.matcher = None
        else:
            raise NotImplementedError
        if self.matcher is not None:
            self.matcher.to(device)
            if self.freeze:
                for p in self.matcher.parameters():
                    p.requires_grad = False
                self.matcher.eval()
            else:
                self.matcher.train()
            if logger is not None:
                n_all = sum(p.numel() for p in self.matcher.parameters())
                logger.log_info(f"Matcher - type: {self.config.type} - freeze: {self.config.freeze} - all_params: {n_all}")
        elif logger is not None:
            logger.log_info(f"Matcher - type: {self.config.type} - freeze: {self.config.freeze}")

    @property
    def _cols(self):
        return 3 if self.matcher_type == "MNN" else 2

    def match_batched(self, bf0, bf1):
        """BatchedFeats x2 -> MatchResult on the device (no sync)."""
        return self.matcher.match_batched(from_batched(bf0), from_batched(bf1))

    def materialize(self, r, n_host, m_host, nmatch_host):
        return materialize_matches(r, n_host, m_host, nmatch_host, self._cols)

    def forward(self, feats0, feats1, *args, **kargs):
        if self.matcher is None:
            return {"matches0": None, "matches1": None, "matching_scores0": None, "matching_scores1": None, "similarity": None,
                    "log_assignment": None}
        if not self.freeze:
            raise NotImplementedError(
                "einx: the trainable matcher branch (random padding to max_points_num) is training-only and out of scope; "
                "set matcher.freeze: true")
        with torch.no_grad():
            pb0, pb1 = from_feats(feats0), from_feats(feats1)
            r = self.matcher.match_batched(pb0, pb1)
            nm = r.nmatch.cpu().tolist()
            n = pb0.counts_host or pb0.counts.cpu().tolist()
            m = pb1.counts_host or pb1.counts.cpu().tolist()
            return self.materialize(r, n, m, nm)
