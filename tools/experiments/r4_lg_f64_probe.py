"""Scratch (build container only): the fp32 reference against the SAME reference in float64 (its own rounding error)."""
import sys, os, copy
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden"))
import gen_golden as g
import torch, numpy as np

for c in g.LG_CASES:
    lg = g.LightGlue(g._ref_stubs.to_attr({"input_dim": c["input_dim"], "ratio_thresh": False, "distance_thresh": False}))
    g.load_synth_weights(lg, c["wseed"]); lg.eval()
    d0, d1, k0, k1 = g.lg_inputs(c)
    size = torch.tensor([260, 346])
    f = lambda d, k, dt: {"sparse_descriptors": torch.from_numpy(d)[None].to(dt), "sparse_positions": torch.from_numpy(k)[None].to(dt), "image_size": [size]}
    with torch.no_grad():
        r32 = lg(f(d0, k0, torch.float32), f(d1, k1, torch.float32))
        lg64 = copy.deepcopy(lg).double()
        r64 = lg64(f(d0, k0, torch.float64), f(d1, k1, torch.float64))
    print(c["name"], "la |ref32-ref64| max", float((r32["log_assignment"].double() - r64["log_assignment"]).abs().max()),
          "ms", float((r32["matching_scores0"].double() - r64["matching_scores0"]).abs().max()),
          "ref_desc", float((r32["ref_descriptors0"].double() - r64["ref_descriptors0"]).abs().max()),
          "flips", int((r32["matches0"] != r64["matches0"]).sum()), r64["log_assignment"].dtype)
