"""Scratch: SP+LightGlue at B=1 in the two workload regimes (independent networks / same scene + calibrated head): wall time and
the per-kernel-class device time of one forward (library-side HIP events)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("ei-nexus_official_amd")
dev = torch.device("cuda", 0)
for same in (False, True):
    for B in (1, 64):
        w = bench.Workload(pkg, dev, "sp_lg", B, same_scene=same)
        sec, mm = w.timed(30 if B == 1 else 5, init=3)
        prof = bench.library_profile(pkg, lambda: (w.step(), torch.cuda.synchronize()))
        top = sorted(prof.items(), key=lambda kv: -kv[1][1])[:8]
        print(f"same_scene={same} B={B}: {sec*1e3:.3f} ms/step, matches {mm:.0f};", ", ".join(f"{k} x{c} {ms:.3f}" for k, (c, ms) in top))
        del w
        torch.cuda.empty_cache()
