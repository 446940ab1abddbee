"""Scratch (build container only): twin scene + centring final_proj (bias = -W mean(x)) + temperature sweep."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden"))
import gen_golden as g
import torch, numpy as np

name = sys.argv[1] if len(sys.argv) > 1 else "sp_lg"
twin = int(sys.argv[2]) if len(sys.argv) > 2 else 1
c = [c for c in g.E2E_CASES if c["name"] == name][0]
cfg = g.model_cfg(c["event_type"], c["image_type"], c["matcher"], c["ce"], 1024, lg_input_dim=(128 if c["image_type"] == "silk" else 256))
model, keys = g.build_eim(cfg, c["wseed"])
sd = model.state_dict()
ev, mask = g.synth.synth_events(c["iseed"], c["B"], c["ce"])
img = g.synth.synth_image(c["iseed"], c["B"])
if twin:
    ek = [k for k in sd if k.startswith("event_extractor.extractor.") and k.endswith("0.weight") and sd[k].dim() == 4]
    ik = [k for k in sd if k.startswith("image_extractor.extractor.") and k.endswith(".weight") and sd[k].dim() == 4]
    new = {}
    for a, b in zip(ek, ik):
        w = sd[b]
        if sd[a].shape[1] != w.shape[1]:
            w = w.repeat(1, sd[a].shape[1], 1, 1) / sd[a].shape[1]
        new[a] = w.clone(); new[a[:-6] + "bias"] = sd[b[:-6] + "bias"].clone()
    for k in sd:
        if k.startswith("event_extractor") and k not in new:
            leaf = k.rsplit(".", 1)[-1]
            if leaf == "running_mean": new[k] = torch.zeros_like(sd[k])
            elif leaf == "running_var": new[k] = torch.ones_like(sd[k])
            elif leaf == "weight" and sd[k].dim() == 1: new[k] = torch.ones_like(sd[k])
            elif leaf == "bias" and k[:-4] + "running_mean" in sd: new[k] = torch.zeros_like(sd[k])
    model.load_state_dict(new, strict=False)
    ev = (ev * np.float32(0.05) + img / np.float32(255.0)).astype(np.float32)
g.calibrate(model, ev, mask, img)
with torch.no_grad():
    ef = model.event_extractor(torch.from_numpy(ev), torch.from_numpy(mask))
    imf = model.image_extractor(torch.from_numpy(img.copy()), None)
lg = model.matcher.matcher
cap = {}
h = lg.transformers[8].register_forward_hook(lambda mod, i, o: cap.__setitem__("x", (o[0].detach(), o[1].detach())))
with torch.no_grad():
    model.matcher(ef, imf)
h.remove()
x0, x1 = cap["x"]
x = torch.cat([x0[0], x1[0]], 0)
print("final desc: norm mean", float(x.norm(dim=1).mean()), "mean-vector norm", float(x.mean(0).norm()), "centred norm mean", float((x - x.mean(0)).norm(dim=1).mean()))
ma = lg.log_assignment[8]
W = ma.final_proj.weight.detach().clone()
sd0 = {k: v.clone() for k, v in lg.state_dict().items()}
xc = x - x.mean(0)
cn = float(xc.norm(dim=1).mean())
zmean = float((ma.matchability(x)).mean())
print("matchability z mean", zmean)
for T in (6, 9, 12, 16):
    # scale so that a centred descriptor of typical norm has |mdesc|^2 = T:  (s*cn)^2/16 = T
    s = (T ** 0.5) * 4.0 / cn
    sdl = {k: v.clone() for k, v in sd0.items()}
    sdl["log_assignment.8.final_proj.weight"] = W * s
    sdl["log_assignment.8.final_proj.bias"] = -(W * s) @ x.mean(0)
    sdl["log_assignment.8.matchability.bias"] = sd0["log_assignment.8.matchability.bias"] - zmean + 3.0
    lg.load_state_dict(sdl)
    with torch.no_grad():
        m = model.matcher(ef, imf)
    ms = m["matching_scores0"][0].reshape(-1)
    m0 = m["matches0"][0].reshape(-1)
    la = m["log_assignment"][0]
    v = ms[m0 > -1].numpy()
    print(f"  T={T} s={s:.3f}: matches {int((m0>-1).sum())}, >0.1: {int((v>0.1).sum())} >0.5 {int((v>0.5).sum())} >0.9 {int((v>0.9).sum())} q10/50/90 {np.quantile(v, [0.1,0.5,0.9]).round(4) if len(v) else None}, |la|max {float(la.abs().max()):.1f}")
