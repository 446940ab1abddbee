"""MatchingRatio / MeanMatchingAccuracy with the reference's class names and `update_one` signatures
(core/metrics/matching_metrics.py:30-51, :84-156), computed by csrc/metrics.hip.  HomographyEstimation
and RelativePoseEstimation stay on the CPU in the reference (cv2 RANSAC) and are out of scope."""
import torch

from ._native_metrics import single_pair


class MatchingRatio:
    def __init__(self, name):
        self.metric_name = name

    def update_one(self, matched_keypoints1, matched_keypoints2, keypoints1, keypoints2):
        assert len(matched_keypoints1) == len(matched_keypoints2)
        r = single_pair(keypoints1, keypoints2, None, None, matched_keypoints1, matched_keypoints2, (1, 1), (1, 1), None, (), ())
        return {self.metric_name: r["MR"]}


class MeanMatchingAccuracy:
    def __init__(self, name, threshold=3, ordering="yx"):
        assert ordering in {"xy", "yx"}
        self.metric_name = name
        self._threshold = threshold
        self._ordering = ordering

    @torch.no_grad()
    def update_one(self, matched_keypoints, warped_matched_keypoints, true_homography):
        assert len(matched_keypoints) == len(warped_matched_keypoints)
        if matched_keypoints.numel() == 0 or warped_matched_keypoints.numel() == 0:
            return {self.metric_name: 0.0}
        r = single_pair(matched_keypoints[:0], warped_matched_keypoints[:0], None, None, matched_keypoints, warped_matched_keypoints,
                        (1, 1), (1, 1), true_homography, (self._threshold,), (), ordering=self._ordering)
        return {self.metric_name: r[f"MMA@{self._threshold}"]}


def compute_auc(errors, thresholds):
    """area under the recall-vs-error curve up to each threshold, normalised (matching_metrics.py:8-27);
    host-side bookkeeping over a list of per-pair errors, numpy like the reference."""
    import numpy as np
    errors = np.array(errors) if isinstance(errors, list) else errors
    errors = errors[np.isfinite(errors)].astype(np.float32)
    errors = np.sort(errors, kind="stable")
    recall = (np.arange(len(errors)) + 1) / len(errors)
    errors = np.r_[0.0, errors]
    recall = np.r_[0.0, recall]
    aucs = {}
    for thres in thresholds:
        last = np.searchsorted(errors, thres)
        rec = np.r_[recall[:last], recall[last - 1]]
        err = np.r_[errors[:last], thres]
        aucs[f"{thres}"] = float(np.sum((err[1:] - err[:-1]) * (rec[1:] + rec[:-1]) * 0.5) / thres)  # np.trapz
    return aucs


class _NeedsOpenCV:
    """HomographyEstimation / RelativePoseEstimation (matching_metrics.py:188-345, :347-470) are cv2 RANSAC
    estimators on the host: downstream of the hot path and out of this build's scope.  The names
    exist so that the evaluation scripts' import lines resolve; constructing one says what is missing."""

    def __init__(self, *a, **k):
        raise NotImplementedError(f"{type(self).__name__} needs OpenCV's RANSAC estimators (cv2.findHomography / cv2.findEssentialMat); "
                                  "it runs on the host after the path and is not part of the native build")


class HomographyEstimation(_NeedsOpenCV):
    pass


class RelativePoseEstimation(_NeedsOpenCV):
    pass
