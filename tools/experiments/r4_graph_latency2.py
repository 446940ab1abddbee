"""Scratch: EIM.forward_graph SP+MNN B=1 wall time (3 x 300 forwards, min of the three means) + phase split."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("ei-nexus_official_amd")
dev = torch.device("cuda", 0)
w = bench.Workload(pkg, dev, "sp_mnn", 1)
fn = lambda: w.model.forward_graph(w.ev, w.img_src, w.mask)
for _ in range(30):
    fn()
best = 1e9
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        fn()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 300 * 1e3)
g = list(w.model._graphs.values())[0]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    g["graph"].replay()
e1.record()
torch.cuda.synchronize()
print(f"forward_graph {best:.3f} ms per pair; graph replay alone (device, back to back) {e0.elapsed_time(e1) / 100:.3f} ms",
      os.environ.get("EINX_CONV_SMALL_PICK_P"), os.environ.get("EINX_CONV_SMALL_PICK_N"))
