"""Scratch: per-forward wall times of 300 SP+LightGlue B=1 forwards; which ones stall?"""
import importlib, os, sys, time, gc
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("ei-nexus_official_amd")
dev = torch.device("cuda", 0)
same = bool(int(sys.argv[1])); nogc = len(sys.argv) > 2 and sys.argv[2] == "nogc"
w = bench.Workload(pkg, dev, "sp_lg", 1, same_scene=same)
if nogc:
    gc.disable()
ts = []
for i in range(300):
    t0 = time.perf_counter()
    w.step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
slow = [(i, round(t, 1)) for i, t in enumerate(ts) if t > 8]
import statistics
print("same_scene", same, "nogc", nogc, "median %.3f ms" % statistics.median(ts), "slow forwards:", slow, "gc counts", gc.get_count(), gc.get_threshold())
