"""ValidDescriptorsDistance with the reference's class name and `update_one` signature
(core/metrics/keypoints_metrics.py:160-290), computed by csrc/metrics.hip."""
import torch

from ._native_metrics import single_pair


class ValidDescriptorsDistance:
    def __init__(self, name, distance_thresh_list, ordering="xy"):
        assert ordering in ["xy", "yx"]
        self.distance_thresh_list = list(distance_thresh_list)
        self.metric_name = name
        self.ordering = ordering

    @torch.no_grad()
    def update_one(self, points1, points2, desc1, desc2, img1_shape, img2_shape, homography):
        assert homography.shape == (3, 3)
        # the reference's default ordering="xy" swaps the first two columns, i.e. it expects (y,x) rows
        kp_order = "yx" if self.ordering == "xy" else "xy"
        r = single_pair(points1, points2, desc1, desc2, None, None, img1_shape, img2_shape, homography, (), tuple(self.distance_thresh_list),
                        ordering=kp_order)
        out = {}
        for t in self.distance_thresh_list:
            for part in ("Repeatability", "ValidDistance", "Angle"):
                out[f"{self.metric_name}_{part}@{t}"] = r[f"VDD_{part}@{t}"]
        return out
