"""Scratch: 60 SP+MNN B=1 forwards (EIM.forward) for a rocprofv3 kernel trace; argv[1] = batch."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("ei-nexus_official_amd")
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w = bench.Workload(pkg, dev, "sp_mnn", B)
for _ in range(20):
    w.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(60):
    w.step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 60 * 1e3)
