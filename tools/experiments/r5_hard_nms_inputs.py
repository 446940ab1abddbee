"""round 5: which image makes the NMS fix-point (radius 3, generic pass kernel) exceed the default pass budget?  (test design aid)"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from helpers import load_pkg, synth
pkg = load_pkg()
DEV = "cuda:0"
cfg = pkg.default_config("SP_MNN", event_channels=5)
cfg.event_extractor.vgg.nms_radius = 3
cfg.image_extractor.superpointv1.nms_radius = 3
ev, mask = synth.synth_events(70, 1, 5)
img = synth.synth_image(70, 1)
H, W = img.shape[-2:]
cands = {"random": img, "const": np.full_like(img, 128.0),
         "xramp": np.broadcast_to(np.linspace(0, 255, W, dtype=np.float32), img.shape).copy(),
         "diag": (np.add.outer(np.arange(H), np.arange(W)) * (255.0 / (H + W))).astype(np.float32)[None, None],
         "quant": np.floor(img / 64) * 64, "blur": None}
k = torch.ones(1, 1, 31, 31) / 961
cands["blur"] = torch.nn.functional.conv2d(torch.from_numpy(img), k, padding=15).numpy()
for name, im in cands.items():
    model = pkg.EIM(cfg, device=DEV).eval()
    sd = synth.synth_state_dict([(k_, tuple(v.shape)) for k_, v in model.state_dict().items()], seed=29)
    model.load_state_dict({k_: torch.from_numpy(v) for k_, v in sd.items()}, strict=False)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32) if a.dtype != bool else a).to(DEV)
    model(t(ev), t(im.astype(np.float32)), t(mask))
    e = model.image_extractor.extractor.engine()
    print(name, "image-side budget", e.nms_iters, "base", e.nms_base, "event side", model.event_extractor.extractor.engine().nms_iters)
