"""Scratch (build container only): synth.twin_overrides + synth.lightglue_calibration through the reference."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden"))
import gen_golden as g
import torch, numpy as np

name = sys.argv[1] if len(sys.argv) > 1 else "sp_lg"
c = [c for c in g.E2E_CASES if c["name"] == name][0]
cfg = g.model_cfg(c["event_type"], c["image_type"], c["matcher"], c["ce"], 1024, lg_input_dim=(128 if c["image_type"] == "silk" else 256))
model, keys = g.build_eim(cfg, c["wseed"])
sd = {k: v.numpy() for k, v in model.state_dict().items()}
model.load_state_dict({k: torch.from_numpy(v) for k, v in g.synth.twin_overrides(sd).items()}, strict=False)
B = 2
ev, mask = g.synth.synth_events(c["iseed"], B, c["ce"])
img = g.synth.synth_image(c["iseed"], B)
ev = g.synth.twin_events(ev, img)
g.calibrate(model, ev, mask, img)
with torch.no_grad():
    ef = model.event_extractor(torch.from_numpy(ev), torch.from_numpy(mask))
    imf = model.image_extractor(torch.from_numpy(img.copy()), None)
for b in range(B):
    d0, d1 = ef["sparse_descriptors"][b], imf["sparse_descriptors"][b]
    S = d0 @ d1.T
    a0, a1 = S.argmax(1), S.argmax(0)
    print("pair", b, "kpts", len(d0), len(d1), "input mutual NN", int((a1[a0] == torch.arange(len(a0))).sum()))
lg = model.matcher.matcher
one = lambda f: {k: f[k][0][None] for k in ("sparse_positions", "sparse_descriptors", "image_size")}
with torch.no_grad():
    r = lg(one(ef), one(imf))
x = np.concatenate([r["ref_descriptors0"][0, 0].numpy(), r["ref_descriptors1"][0, 0].numpy()], 0)
print(x.shape)
lg = model.matcher.matcher
lsd = {k: v.numpy().copy() for k, v in lg.state_dict().items()}
for T in (32, 48, 64, 80, 100):
    over, s = g.synth.lightglue_calibration(lsd, x, temperature=T)
    lg.load_state_dict({k: torch.from_numpy(v) for k, v in over.items()}, strict=False)
    with torch.no_grad():
        m = model.matcher(ef, imf)
    for b in range(B):
        ms = m["matching_scores0"][b].reshape(-1); m0 = m["matches0"][b].reshape(-1); la = m["log_assignment"][b]
        v = ms[m0 > -1].numpy()
        print(f"  T={T} s={s:.3f} pair {b}: matches {int((m0>-1).sum())}, >0.1: {int((v>0.1).sum())} >0.5 {int((v>0.5).sum())} >0.9 {int((v>0.9).sum())} q10/50/90 {np.quantile(v, [0.1,0.5,0.9]).round(4)}, |la|max {float(la.abs().max()):.1f}")
