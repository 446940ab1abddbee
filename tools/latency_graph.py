"""Single-pair wall time through EIM.forward_graph (one hipGraph launch per forward) next to EIM.forward: SP+MNN and
SP+LightGlue, 3 x 300 forwards each (the minimum of the three means), and the graph replayed back to back (device time)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("ei-nexus_official_amd")
dev = torch.device("cuda", 0)


def best_of(fn, reps=3, n=300):
    for _ in range(30):
        fn()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best


for cfg in ("sp_mnn", "sp_lg"):
    w = bench.Workload(pkg, dev, cfg, 1)
    eager = best_of(lambda: (w.img.copy_(w.img_src), w.model(w.ev, w.img, w.mask)), n=300 if cfg == "sp_mnn" else 100)
    graph = best_of(lambda: w.model.forward_graph(w.ev, w.img_src, w.mask), n=300 if cfg == "sp_mnn" else 100)
    g = list(w.model._graphs.values())[0]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        g["graph"].replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"B=1 {cfg}: EIM.forward {eager:.3f} ms per pair, EIM.forward_graph {graph:.3f} ms per pair, graph replayed back to back "
          f"{e0.elapsed_time(e1) / 100:.3f} ms (device)")
    del w
