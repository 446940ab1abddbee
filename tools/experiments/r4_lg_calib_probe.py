"""Scratch (build container only): how do the LightGlue e2e fixtures look, and what final_proj scale gives confident matches?"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden"))
import gen_golden as g
import torch, numpy as np

name = sys.argv[1] if len(sys.argv) > 1 else "sp_lg"
c = [c for c in g.E2E_CASES if c["name"] == name][0]
cfg = g.model_cfg(c["event_type"], c["image_type"], c["matcher"], c["ce"], 1024, lg_input_dim=(128 if c["image_type"] == "silk" else 256))
model, keys = g.build_eim(cfg, c["wseed"])
ev, mask = g.synth.synth_events(c["iseed"], c["B"], c["ce"])
img = g.synth.synth_image(c["iseed"], c["B"])
g.calibrate(model, ev, mask, img)
t = time.time()
with torch.no_grad():
    ef = model.event_extractor(torch.from_numpy(ev), torch.from_numpy(mask))
    imf = model.image_extractor(torch.from_numpy(img.copy()), None)
print("extract", time.time() - t)
lg = model.matcher.matcher
sd0 = {k: v.clone() for k, v in lg.state_dict().items()}
for s, zb in [(1, 0), (2, 3), (3, 3), (4, 3), (6, 3), (8, 5)]:
    sd = {k: v.clone() for k, v in sd0.items()}
    sd["log_assignment.8.final_proj.weight"] *= s
    sd["log_assignment.8.final_proj.bias"] *= s
    sd["log_assignment.8.matchability.bias"] += zb
    lg.load_state_dict(sd)
    t = time.time()
    with torch.no_grad():
        m = model.matcher(ef, imf)
    ms = m["matching_scores0"][0].reshape(-1)
    m0 = m["matches0"][0].reshape(-1)
    la = m["log_assignment"][0]
    v = ms[m0 > -1]
    print(f"s={s} zb={zb}: matches {int((m0>-1).sum())}, scores q10/50/90 {np.quantile(v.numpy(), [0.1,0.5,0.9]) if len(v) else None}, |la|max {float(la.abs().max()):.1f}, t {time.time()-t:.1f}s")
