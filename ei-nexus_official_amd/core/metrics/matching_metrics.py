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
