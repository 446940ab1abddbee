"""Matcher wrapper (reference core/modules/Matchers.py:13-222): config dispatch, the frozen
inference branch and the forward pass of the un-frozen branch.  The reference calls its frozen
matcher once per sample; here the whole batch goes to the device in one launch sequence and the
per-sample lists are cut afterwards.  The un-frozen branch (Matchers.py:204-222)
pads every sample to max_points_num with random keypoints / descriptors, stacks the batch and makes
ONE batched matcher call; its forward values are reproduced here (the random draws come from the
same torch generators in the same order), autograd and the losses are not part of this build."""
import torch
from torch import nn

from ..._native import on_input_device
from ... import _native as N

from .matchers.MNN import NearestNeighborMatcher
from .matchers.lightglue import LightGlue
from .matchers._batched import from_batched, from_feats, full_batch_lists, materialize_matches


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

    @on_input_device
    def match_batched(self, bf0, bf1):
        """BatchedFeats x2 -> MatchResult on the device (no sync).  The matched keypoints are also
        packed pair after pair so that materialize() cuts the per-pair lists with one split call."""
        return N.compact_matches(self.matcher.match_batched(from_batched(bf0), from_batched(bf1)))

    def materialize(self, r, n_host, m_host, nmatch_host, prebuilt=None):
        return materialize_matches(r, n_host, m_host, nmatch_host, self._cols, prebuilt=prebuilt)

    # ---- un-frozen branch: pad to max_points_num (Matchers.py:67-149) ---------------------------
    def pad_sparse_positions_to_length(self, sparse_positions, length, image_size=None):
        """[N,3] -> [length,3].  'random': uniform positions inside the image with score 0, drawn with
        torch.rand on the keypoints' device exactly as Matchers.py:80-91 does (same generator, same
        shape, same order => same stream of draws); the scaling runs in einx_random_positions."""
        n = len(sparse_positions)
        if n < length:
            r = length - n
            dev = sparse_positions.device
            if self.pad_mode == "zeros":
                pad = torch.zeros(r, 3, device=dev)
            elif self.pad_mode == "random":
                if image_size is None:
                    image_size = sparse_positions[:, 0].max(), sparse_positions[:, 1].max()
                if isinstance(image_size, list):
                    image_size = image_size[0]
                u = torch.rand(r, 2, device=dev)
                pad = N.random_positions(u, (float(image_size[0]), float(image_size[1])))
            else:
                raise NotImplementedError(f"Unknown mode: {self.pad_mode}")
            sparse_positions = torch.cat([sparse_positions, pad], dim=0)
        elif n > length:
            sparse_positions = sparse_positions[:length, ...]
        return sparse_positions

    def pad_sparse_descriptors_to_length(self, sparse_descriptors, length):
        """[N,C] -> [length,C].  'random': torch.randn on the CPU generator (Matchers.py:114-117), then
        F.normalize * desc_scale_factor on the device (einx_normalize_rows)."""
        n = len(sparse_descriptors)
        if n < length:
            r = length - n
            dev, C = sparse_descriptors.device, sparse_descriptors.shape[1]
            if self.pad_mode == "zeros":
                pad = torch.zeros(r, C, device=dev)
            elif self.pad_mode == "random":
                pad = N.normalize_rows(torch.randn(r, C).to(dev), float(self.desc_scale_factor))
            else:
                raise NotImplementedError(f"Unknown mode: {self.pad_mode}")
            sparse_descriptors = torch.cat([sparse_descriptors, pad], dim=0)
        elif n > length:
            sparse_descriptors = sparse_descriptors[:length, ...]
        return sparse_descriptors

    def pad_sparse_feats_to_length(self, feats, length):
        pos, desc = feats["sparse_positions"], feats["sparse_descriptors"]
        image_size = feats["image_size"][::-1]  # list reversed (not (H,W) swapped), as Matchers.py:135
        out_pos, out_desc = [], []
        for i in range(len(pos)):
            out_pos.append(self.pad_sparse_positions_to_length(pos[i], length, image_size))
            out_desc.append(self.pad_sparse_descriptors_to_length(desc[i], length))
        feats["sparse_positions"] = out_pos
        feats["sparse_descriptors"] = out_desc
        return feats

    def stack_sparse_feats(self, feats):
        feats["sparse_positions"] = torch.stack(feats["sparse_positions"], dim=0)
        feats["sparse_descriptors"] = torch.stack(feats["sparse_descriptors"], dim=0)
        return feats

    @on_input_device
    def forward(self, feats0, feats1, *args, **kargs):
        if self.matcher is None:
            return {"matches0": None, "matches1": None, "matching_scores0": None, "matching_scores1": None, "similarity": None,
                    "log_assignment": None}
        if not self.freeze:
            feats0 = self.stack_sparse_feats(self.pad_sparse_feats_to_length(feats0, self.max_points_num))
            feats1 = self.stack_sparse_feats(self.pad_sparse_feats_to_length(feats1, self.max_points_num))
            matches = self.matcher(feats0, feats1)
            matches["input_feats0"] = feats0
            matches["input_feats1"] = feats1
            return matches
        with torch.no_grad():
            pb0, pb1 = from_feats(feats0), from_feats(feats1)
            r = self.matcher.match_batched(pb0, pb1)
            if getattr(r, "stale", None) is not None:  # LightGlue: match counts + the weight watch in one read-back
                nm = torch.cat([r.nmatch, r.stale]).cpu().tolist()
                if nm.pop():  # a weight was edited through `.data`: rebuild the folded images, match again
                    self.matcher.refresh()
                    r = self.matcher.match_batched(pb0, pb1)
                    nm = r.nmatch.cpu().tolist()
            else:
                nm = r.nmatch.cpu().tolist()
            n = pb0.counts_host or pb0.counts.cpu().tolist()
            m = pb1.counts_host or pb1.counts.cpu().tolist()
            return self.materialize(r, n, m, nm)
