"""Repeatability and ValidDescriptorsDistance with the reference's class names and `update_one` /
`update_batch` signatures (core/metrics/keypoints_metrics.py:54-157, :160-329), computed by csrc/metrics.hip."""
import torch

from ._native_metrics import single_pair


class Repeatability:
    """(count1 + count2) / (N + M) of mutual nearest warped keypoints within `distance_thresh`
    (keypoints_metrics.py:54-157).  The same neighbour pass as ValidDescriptorsDistance's repeatability
    (identical whenever either image keeps a keypoint after the visibility filter; with none on both sides the reference
    emits no entry -- the kernel marks that case with NaN (`rep_nan_if_empty`) and the entry is dropped here).
    ordering "xy": rows are (x, y); "yx": rows are (y, x) -- note the opposite convention of
    ValidDescriptorsDistance (keypoints_metrics.py:86-91 vs :193-198)."""

    def __init__(self, name, distance_thresh=3, ordering="xy"):
        assert ordering in ["xy", "yx"]
        self.distance_thresh = distance_thresh
        self.metric_name = name
        self.ordering = ordering

    @torch.no_grad()
    def update_one(self, points1, points2, img1_shape, img2_shape, homography):
        assert homography.shape == (3, 3)
        if points1.shape[0] + points2.shape[0] == 0:
            return {}
        r = single_pair(points1, points2, None, None, None, None, img1_shape, img2_shape, homography, (), (self.distance_thresh,),
                        ordering=self.ordering, rep_nan_if_empty=True)
        v = r[f"VDD_Repeatability@{self.distance_thresh}"]
        return {} if v != v else {self.metric_name: v}  # NaN: original_num + warped_num == 0 (keypoints_metrics.py:126)

    @torch.no_grad()
    def update_batch(self, points1, points2, img1_shape, img2_shape, homography):
        assert len(points1) == len(points2) == len(homography)
        vals = []
        for i in range(len(points1)):
            one = self.update_one(points1[i], points2[i], img1_shape, img2_shape, homography[i])
            if self.metric_name in one:
                vals.append(one[self.metric_name])
        return {self.metric_name: torch.tensor(vals).mean().item()}


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
