"""Scratch: full per-kernel-class device time of one same-scene SP+LightGlue B=1 forward."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("ei-nexus_official_amd")
dev = torch.device("cuda", 0)
for same in (True, False):
    w = bench.Workload(pkg, dev, "sp_lg", 1, same_scene=same)
    for _ in range(5):
        w.step()
    prof = bench.library_profile(pkg, lambda: (w.step(), torch.cuda.synchronize()))
    print("same_scene", same, "sum ms", round(sum(ms for c, ms in prof.values()), 3))
    for k, (c, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k} x{c} {ms:.3f}")
