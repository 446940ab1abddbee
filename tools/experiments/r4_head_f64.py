"""Scratch (build container only): how much of the log_assignment error is the assignment head (final_proj, sim, softmaxes)
and how much the nine layers before it?  float64 head on the fp32 reference's final descriptors vs the all-float64 reference."""
import sys, os, copy
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden"))
import gen_golden as g
import torch, numpy as np

def run(lg, f0, f1, tag):
    cap = {}
    h = lg.transformers[8].register_forward_hook(lambda mod, i, o: cap.__setitem__("x", (o[0].detach(), o[1].detach())))
    with torch.no_grad():
        r32 = lg(f0, f1)
    h.remove()
    x0, x1 = cap["x"]
    lg64 = copy.deepcopy(lg).double()
    dbl = lambda f: {k: (v.double() if torch.is_tensor(v) else v) for k, v in f.items()}
    h = lg64.transformers[8].register_forward_hook(lambda mod, i, o: cap.__setitem__("x64", (o[0].detach(), o[1].detach())))
    with torch.no_grad():
        r64 = lg64(dbl(f0), dbl(f1))
        h.remove()
        la_h, _ = lg64.log_assignment[8](x0.double(), x1.double())      # f64 head on fp32 x
        la_32h, _ = lg.log_assignment[8](cap["x64"][0].float(), cap["x64"][1].float())  # fp32 head on (rounded) exact x
    t = r64["log_assignment"]
    print(tag, "ref32-f64", float((r32["log_assignment"].double() - t).abs().max()), "| f64 head on fp32 x", float((la_h - t).abs().max()),
          "| fp32 head on exact x", float((la_32h.double() - t).abs().max()), "| x err", float((x0.double() - cap["x64"][0]).abs().max()))

for c in g.LG_CASES[:2]:
    lg = g.LightGlue(g._ref_stubs.to_attr({"input_dim": c["input_dim"], "ratio_thresh": False, "distance_thresh": False}))
    g.load_synth_weights(lg, c["wseed"]); lg.eval()
    d0, d1, k0, k1 = g.lg_inputs(c)
    size = torch.tensor([260, 346])
    f = lambda d, k: {"sparse_descriptors": torch.from_numpy(d)[None], "sparse_positions": torch.from_numpy(k)[None], "image_size": [size]}
    run(lg, f(d0, k0), f(d1, k1), c["name"])
c = g.LGCAL_CASES[0]
cfg = g.model_cfg(c["event_type"], c["image_type"], c["matcher"], c["ce"], 1024, lg_input_dim=256)
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
over, scale = g.synth.lightglue_calibration({k: v.numpy().copy() for k, v in lg.state_dict().items()}, x, temperature=32.0)
lg.load_state_dict({k: torch.from_numpy(v) for k, v in over.items()}, strict=False)
run(lg, g._one(ef, 0), g._one(imf, 0), "sp_lg_twin")
