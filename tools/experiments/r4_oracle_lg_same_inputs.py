"""Scratch (build container only): the oracle's LightGlue fed with the REFERENCE's extractor outputs (identical inputs), against
the reference in fp32 / float64; and the reference's sensitivity to 2e-6 input noise."""
import sys, os, copy
ROOT = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, ROOT)
import gen_golden as g
import torch, numpy as np
from oracle import oracle as orc

c = g.LGCAL_CASES[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
cfg = g.model_cfg(c["event_type"], c["image_type"], c["matcher"], c["ce"], 1024, lg_input_dim=(128 if c["image_type"] == "silk" else 256))
model, keys = g.build_eim(cfg, c["wseed"])
sd = {k: v.numpy().copy() for k, v in model.state_dict().items()}
model.load_state_dict({k: torch.from_numpy(v) for k, v in g.synth.twin_overrides(sd).items()}, strict=False)
ev, mask = g.synth.synth_events(c["iseed"], 1, c["ce"]); img = g.synth.synth_image(c["iseed"], 1); ev = g.synth.twin_events(ev, img)
g.calibrate(model, ev, mask, img)
lg = model.matcher.matcher
with torch.no_grad():
    ef = model.event_extractor(torch.from_numpy(ev), torch.from_numpy(mask)); imf = model.image_extractor(torch.from_numpy(img.copy()), None)
    r = lg(g._one(ef, 0), g._one(imf, 0))
x = np.concatenate([r["ref_descriptors0"][0, 0].numpy(), r["ref_descriptors1"][0, 0].numpy()], 0)
over, scale = g.synth.lightglue_calibration({k: v.numpy().copy() for k, v in lg.state_dict().items()}, x)
lg.load_state_dict({k: torch.from_numpy(v) for k, v in over.items()}, strict=False)
f0, f1 = g._one(ef, 0), g._one(imf, 0)
dbl = lambda f: {k: (v.double() if torch.is_tensor(v) else v) for k, v in f.items()}
with torch.no_grad():
    r32 = lg(f0, f1)
    r64 = copy.deepcopy(lg).double()(dbl(f0), dbl(f1))
    noise = lambda t, s: t + torch.from_numpy(g.synth.uniform(s, tuple(t.shape), -2e-6, 2e-6))
    g0 = dict(f0, sparse_descriptors=noise(f0["sparse_descriptors"], 5)); g1 = dict(f1, sparse_descriptors=noise(f1["sparse_descriptors"], 6))
    rn = lg(g0, g1)
lsd = {k: v.numpy() for k, v in lg.state_dict().items()}
ro = orc.lightglue(lsd, f0["sparse_positions"][0].numpy(), f0["sparse_descriptors"][0].numpy(), f1["sparse_positions"][0].numpy(), f1["sparse_descriptors"][0].numpy())
t = r64["log_assignment"][0].numpy()
la_o = ro["log_assignment"].astype(np.float64)
print(c["name"], "same inputs: oracle-f64", np.abs(la_o - t).max(), "oracle-ref32", np.abs(la_o - r32["log_assignment"][0].numpy()).max(), "ref32-f64", np.abs(r32["log_assignment"][0].numpy() - t).max(),
      "| ref32 with 2e-6 input noise vs ref32", float((rn["log_assignment"] - r32["log_assignment"]).abs().max()),
      "| flips oracle/ref32", int((ro["matches0"] != r32["matches0"][0].numpy()).sum()), "ms oracle-f64", np.abs(ro["matching_scores0"] - r64["matching_scores0"][0].numpy()).max())
