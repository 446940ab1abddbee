"""Scratch: what triggers the one-off stalls?  SP+LightGlue B=1, independent regime, variants of the warm-up."""
import importlib, os, sys, time, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("ei-nexus_official_amd")
dev = torch.device("cuda", 0)
mode = sys.argv[1]
w = bench.Workload(pkg, dev, "sp_lg", 1, same_scene=False)
if mode == "reload":      # parent-level load of the SAME weights (drops and rebuilds every native image)
    w.step()
    w.model.load_state_dict({k: torch.from_numpy(v) for k, v in w.sd.items()}, strict=False)
elif mode == "reload_lg":  # only the matcher's weights
    w.step()
    w.model.load_state_dict({k: torch.from_numpy(v) for k, v in w.sd.items() if k.startswith("matcher")}, strict=False)
elif mode == "inner":      # the inner LightGlue module called directly once (what the calibration does)
    ef, imf, _ = w.step()
    one = lambda f: {"sparse_positions": f["sparse_positions"][0][None], "sparse_descriptors": f["sparse_descriptors"][0][None], "image_size": [f["image_size"][0]]}
    w.model.matcher.matcher(one(ef), one(imf))
ts = []
for i in range(60):
    t0 = time.perf_counter()
    w.step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("mode", mode, "median %.3f" % statistics.median(ts), "slow:", [(i, round(t, 1)) for i, t in enumerate(ts) if t > 8])
