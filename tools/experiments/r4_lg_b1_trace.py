"""Scratch: 40 same-scene (argv[1]=1) or independent (0) SP+LightGlue B=1 forwards for a rocprofv3 kernel trace."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("ei-nexus_official_amd")
dev = torch.device("cuda", 0)
w = bench.Workload(pkg, dev, "sp_lg", 1, same_scene=bool(int(sys.argv[1])))
for _ in range(5):
    w.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40):
    w.step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 40 * 1e3)
