"""Scratch (build container only): 'twin scene' recipe -- event extractor = image extractor's weights, events = image/255 + alpha * noise."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden"))
import gen_golden as g
import torch, numpy as np

name = sys.argv[1] if len(sys.argv) > 1 else "sp_lg"
c = [c for c in g.E2E_CASES if c["name"] == name][0]
cfg = g.model_cfg(c["event_type"], c["image_type"], c["matcher"], c["ce"], 1024, lg_input_dim=(128 if c["image_type"] == "silk" else 256))
model, keys = g.build_eim(cfg, c["wseed"])
sd = model.state_dict()
ek = [k for k in sd if k.startswith("event_extractor.extractor.") and k.endswith("0.weight") and sd[k].dim() == 4]
ik = [k for k in sd if k.startswith("image_extractor.extractor.") and k.endswith(".weight") and sd[k].dim() == 4]
print(len(ek), len(ik))
new = {}
for a, b in zip(ek, ik):
    w = sd[b]
    if sd[a].shape[1] != w.shape[1]:
        w = w.repeat(1, sd[a].shape[1], 1, 1) / sd[a].shape[1]
    assert sd[a].shape == w.shape, (a, b)
    new[a] = w.clone(); new[a[:-6] + "bias"] = sd[b[:-6] + "bias"].clone()
for k in sd:
    if k.startswith("event_extractor") and k not in new:
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "running_mean": new[k] = torch.zeros_like(sd[k])
        elif leaf == "running_var": new[k] = torch.ones_like(sd[k])
        elif leaf == "weight" and sd[k].dim() == 1: new[k] = torch.ones_like(sd[k])
        elif leaf == "bias" and k[:-4] + "running_mean" in sd: new[k] = torch.zeros_like(sd[k])
model.load_state_dict(new, strict=False)
ev, mask = g.synth.synth_events(c["iseed"], c["B"], c["ce"])
img = g.synth.synth_image(c["iseed"], c["B"])
lg = model.matcher.matcher
sd0 = {k: v.clone() for k, v in lg.state_dict().items()}
for alpha in (0.02, 0.05, 0.1):
    ev2 = (ev * np.float32(alpha) + img / np.float32(255.0)).astype(np.float32)
    g.calibrate(model, ev2, mask, img)
    with torch.no_grad():
        ef = model.event_extractor(torch.from_numpy(ev2), torch.from_numpy(mask))
        imf = model.image_extractor(torch.from_numpy(img.copy()), None)
    d0, d1 = ef["sparse_descriptors"][0], imf["sparse_descriptors"][0]
    S = d0 @ d1.T
    a0, a1 = S.argmax(1), S.argmax(0)
    print(f"alpha {alpha}: kpts {len(d0)} {len(d1)} input mutual NN", int((a1[a0] == torch.arange(len(a0))).sum()), "sim max", float(S.max()))
    for r, s, zb in [(1.0, 1, 0), (1.0, 4, 4), (0.3, 4, 4), (0.3, 8, 4), (0.1, 8, 4), (0.1, 16, 4)]:
        sdl = {k: v.clone() for k, v in sd0.items()}
        for k in sdl:
            if k.endswith("ffn.3.weight") or k.endswith("ffn.3.bias"):
                sdl[k] *= r
        sdl["log_assignment.8.final_proj.weight"] *= s
        sdl["log_assignment.8.final_proj.bias"] *= s
        sdl["log_assignment.8.matchability.bias"] += zb
        lg.load_state_dict(sdl)
        with torch.no_grad():
            m = model.matcher(ef, imf)
        ms = m["matching_scores0"][0].reshape(-1)
        m0 = m["matches0"][0].reshape(-1)
        la = m["log_assignment"][0]
        v = ms[m0 > -1].numpy()
        print(f"  r={r} s={s} zb={zb}: matches {int((m0>-1).sum())}, >0.1: {int((v>0.1).sum())} q10/50/90 {np.quantile(v, [0.1,0.5,0.9]).round(4) if len(v) else None}, |la|max {float(la.abs().max()):.1f}")
