"""Model configuration defaults with the reference's schema (configs/model/SP_MNN.yaml,
SP_LG.yaml, SiLK_MNN.yaml, test/EI_SiLK_LG.yaml).  Any attribute-style mapping works as a config
(an omegaconf DictConfig if installed, or the AttrDict below)."""
import copy


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(o):
    if isinstance(o, dict):
        return AttrDict({k: to_attr(v) for k, v in o.items()})
    if isinstance(o, (list, tuple)):
        return [to_attr(v) for v in o]
    return o


_EXTRACT = dict(nms_radius=4, detection_threshold=1.0, detection_top_k=1024, remove_borders=4, ordering="yx",
                learnable_descriptor_scale_factor=False)

_BASE = {
    "name": "EIM",
    "pretrain_stage1": {"model_path": None},
    "pretrain_stage2": {"model_path": None},
    "event_extractor": {
        "type": "vgg", "freeze": True,
        "vgg": dict(_EXTRACT, in_channels=16, feat_channels=128, descriptor_dim=256, descriptor_scale_factor=1.0, use_batchnorm=True),
        "vgg_np": dict(_EXTRACT, in_channels=16, feat_channels=128, descriptor_dim=128, descriptor_scale_factor=1.41, use_batchnorm=True,
                       padding=1),
    },
    "image_extractor": {
        "type": "superpointv1", "freeze": True,
        "superpointv1": dict(_EXTRACT, descriptor_dim=256, descriptor_scale_factor=1.0),
        "silk": dict(padding=1, nms_radius=4, detection_threshold=1.0, detection_top_k=1024, remove_borders=4,
                     descriptor_scale_factor=1.41, learnable_descriptor_scale_factor=False),
    },
    "matcher": {
        "type": "MNN", "freeze": True, "max_points_num": 1024, "pad_mode": "random", "desc_scale_factor": 1.0,
        "MNN": {"ratio_thresh": False, "distance_thresh": False},
        "LightGlue": {"ratio_thresh": False, "distance_thresh": False},
    },
}


def default_config(name="SP_MNN", event_channels=5):
    """name in {SP_MNN, SP_LG, SiLK_MNN, SiLK_LG}; event_channels = voxel-grid bins (5 in
    BASELINE.json, 16 in the reference's shipped YAMLs)."""
    cfg = copy.deepcopy(_BASE)
    cfg["event_extractor"]["vgg"]["in_channels"] = event_channels
    cfg["event_extractor"]["vgg_np"]["in_channels"] = event_channels
    if name.startswith("SiLK"):
        cfg["event_extractor"]["type"] = "vgg_np"
        cfg["image_extractor"]["type"] = "silk"
    elif not name.startswith("SP"):
        raise ValueError(name)
    if name.endswith("_LG"):
        cfg["matcher"]["type"] = "LightGlue"
        cfg["matcher"]["LightGlue"]["input_dim"] = 128 if name.startswith("SiLK") else 256
    elif not name.endswith("_MNN"):
        raise ValueError(name)
    return to_attr(cfg)
