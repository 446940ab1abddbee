"""Scratch: cProfile of the eager single-pair EIM.forward (SP+MNN, B=1): which host calls the 0.7-0.8 ms are made of."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("ei-nexus_official_amd")
dev = torch.device("cuda", 0)
w = bench.Workload(pkg, dev, "sp_mnn", 1)
m = w.model


def step():
    w.img.copy_(w.img_src)
    return m(w.ev, w.img, w.mask)


for _ in range(50):
    step()
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(300):
        step()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 300 * 1e3)
print(f"eager forward {best:.3f} ms")
# enqueue only (no waiting): how long the host needs to put one forward on the streams
t0 = time.perf_counter()
for _ in range(300):
    w.img.copy_(w.img_src)
    p = m._enqueue(w.ev, w.img, w.mask)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"_enqueue alone (host) {(t1 - t0) / 300 * 1e3:.3f} ms")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
