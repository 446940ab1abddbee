"""Evaluation harnesses with the call patterns of the reference's test_events-image_same-time.py:130-283
 and test_events-image_different_time.py:187-264, entirely on the device:

    raw events --(events.hip)--> voxel grid + events mask
               --(EIM: conv/detect/desc/mnn|lightglue kernels)--> keypoints, descriptors, matches
               --(metrics.hip)--> MR, MMA@1/3, VDD@1/3 per pair  --> means (all-reduced across ranks)

HomographyEstimation / RelativePoseEstimation (cv2 RANSAC on the CPU in the reference) are out of scope;
DifferentTimeEvaluator hands over exactly what the reference passes to them.
"""
import torch

from .core.metrics._native_metrics import batch_metrics, metric_names
from .datasets.representations import EventStage, events_representation_batch


class SameTimeEvaluator:
    def __init__(self, model, bins, resolution=(346, 260), mma_thr=(1, 3), vdd_thr=(1, 3)):
        """model: EIM (eval mode); bins: voxel-grid channels; resolution: (W, H) like MVSECDataset.RESOLUTION."""
        self.model = model
        self.bins = int(bins)
        self.resolution = tuple(int(v) for v in resolution)
        self.mma_thr, self.vdd_thr = tuple(mma_thr), tuple(vdd_thr)
        self.names = metric_names(self.mma_thr, self.vdd_thr)
        self._sums = None
        self._counts = None
        self._rows = []
        self.pairs = 0

    @torch.no_grad()
    def step(self, events_list, images, homography=None):
        """events_list: B dicts {"x","y","t","p"} of numpy arrays; images: [B,1,H,W] float (0..255) on the device
        (scaled in place by SuperPoint exactly like the reference).  Returns the per-pair metric rows [B,K] (device)."""
        W, H = self.resolution
        dev = images.device
        if not hasattr(self, "_stages"):
            self._stages = {}
        with torch.cuda.device(dev):
            stage = self._stages.get(("step", dev))
            if stage is None:  # page-locked upload path (the call is synchronous: the stage is free again when it returns)
                stage = self._stages[("step", dev)] = EventStage(dev)
            # the image network does not depend on the events: it is enqueued FIRST and the host packs / uploads the raw events
            # (38 MB, ~2.5 ms) under its ~4 ms of device work; the event network follows (round 6: 11.7 -> see profiles/r06_notes.md)
            im = self.model.enqueue_image(images, None)
            events_rep, events_mask = events_representation_batch(events_list, (self.bins, H, W), normalize=True, device=dev, stage=stage)
            self.last_inputs = (events_rep, events_mask)  # what the extractors saw (deterministic since round 4: bit-equal run to run)
            ef, imf, matches = self.model._finish(self.model._enqueue(events_rep, images, events_mask, image_feats=im))
        return self._account(ef, imf, matches, homography)

    def _account(self, ef, imf, matches, homography):
        rows = batch_metrics(ef._batched, imf._batched, self.model._last_match, homography, self.mma_thr, self.vdd_thr)
        # the running sums are folded lazily (`_fold`): eight small reductions per batch on the forward's stream were a third of
        # what the evaluation loop cost on top of the forward (profiles/r06_notes.md 3)
        self._rows.append(rows)
        if len(self._rows) >= 64:
            self._fold()
        self.pairs += rows.shape[0]
        return rows, (ef, imf, matches)

    def _fold(self):
        if not self._rows:
            return
        rows = torch.cat(self._rows, 0) if len(self._rows) > 1 else self._rows[0]
        self._rows = []
        ok = ~torch.isnan(rows)
        z = torch.nan_to_num(rows)
        self._sums = z.sum(0) if self._sums is None else self._sums + z.sum(0)
        self._counts = ok.sum(0).double() if self._counts is None else self._counts + ok.sum(0).double()

    @property
    def sums(self):
        self._fold()
        return self._sums

    @property
    def counts(self):
        self._fold()
        return self._counts

    @torch.no_grad()
    def run(self, batches, depth=2):
        """The evaluation LOOP (test_events-image_same-time.py:130-194 iterates a DataLoader): `batches` yields
        (events_list, images[, homography]) like the arguments of `step`; one `step` result per batch comes back, in order.
        Up to `depth` batches are in flight: batch i + 1's events are concatenated into page-locked memory, uploaded with
        non-blocking copies and its kernels enqueued (EIM.forward_stream's mechanism) BEFORE the host waits for batch i's
        counts, so packing and the PCIe transfer hide under the device's work instead of adding to it.  Same kernels, same
        results as `step`; every `images` tensor must stay untouched until its result has been yielded."""
        from collections import deque
        W, H = self.resolution
        depth = max(int(depth), 1)
        pending = deque()
        if not hasattr(self, "_stages"):
            self._stages = {}
        k = 0

        def finish(entry):
            p, hom = entry
            return self._account(*self.model._finish(p), hom)

        for item in batches:
            events_list, images = item[0], item[1]
            homography = item[2] if len(item) > 2 else None
            slot = k % depth
            dev = images.device
            with torch.cuda.device(dev):
                stage = self._stages.get((slot, dev))
                if stage is None:
                    stage = self._stages[(slot, dev)] = EventStage(dev)
                # upload AND representation kernels of batch i + 1 on the stage's own stream: they run beside batch i's
                # convolutions (0.3 ms of memory- / latency-bound kernels per batch leave the main stream's chain).  (Every
                # in-flight slot on a stream of its own, so that batch i's tail could overlap batch i + 1's head, measured the
                # same: 8.89 vs 8.84 ms per batch, profiles/r06_notes.md.)
                rep, mask = events_representation_batch(events_list, (self.bins, H, W), normalize=True, device=dev, stage=stage, on_stage_stream=True)
                self.last_inputs = (rep, mask)  # of the batch enqueued last (results lag by up to depth - 1 batches)
                pending.append((self.model._enqueue(rep, images, mask, slot=slot), homography))
            k += 1
            if len(pending) >= depth:
                yield finish(pending.popleft())
        while pending:
            yield finish(pending.popleft())

    def result(self):
        """Mean of every metric over the pairs seen so far; sums are all-reduced when a process group is up."""
        s, c = self.sums.clone(), self.counts.clone()
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(s)
            torch.distributed.all_reduce(c)
        mean = (s / c.clamp_min(1)).tolist()
        return dict(zip(self.names, mean))


class DifferentTimeEvaluator(SameTimeEvaluator):
    """Call pattern of test_events-image_different_time.py:187-264: the events come from frame i, the image from a LATER
    frame j of the sequence, and the two views are related by a known motion instead of the identity.

    * `step(events_list, images, homography)`: as SameTimeEvaluator.step, with the image-0 -> image-1 homography [B,3,3]
      of the pair (planar scenes / pure rotations; None = identity) going into MMA@t and VDD@t (metrics.hip warps the
      keypoints exactly like core/metrics/util.py:warp_points).
    * `pose_inputs(matches, b)`: what the reference feeds RelativePoseEstimation.update_one for pair b
      (`matches["matched_kpts0"][b]`, `matches["matched_kpts1"][b]`, :251-257), plus the (x, y) views of
      test_events-image_different_time.py:217-224 (`[..., :2]`, flipped when the extractor's ordering is "yx").  The pose
      solver itself (cv2.findEssentialMat / recoverPose on the CPU) is outside this build.
    """

    def pose_inputs(self, matches, b=0):
        mk0, mk1 = matches["matched_kpts0"][b], matches["matched_kpts1"][b]
        xy0, xy1 = mk0[..., :2], mk1[..., :2]
        if self.model.event_extractor.extractor.ordering == "yx":
            xy0, xy1 = torch.flip(xy0, dims=[-1]), torch.flip(xy1, dims=[-1])
        return {"matched_kpts0": mk0, "matched_kpts1": mk1, "matched_xy0": xy0, "matched_xy1": xy1}
