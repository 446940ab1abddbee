"""Scratch: single-pair wall time of EIM.forward vs EIM.forward_graph (SP+MNN and SP+LightGlue)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("ei-nexus_official_amd")
dev = torch.device("cuda", 0)
for cfg in ("sp_mnn", "sp_lg"):
    w = bench.Workload(pkg, dev, cfg, 1)
    for name, fn in (("forward", lambda: (w.img.copy_(w.img_src), w.model(w.ev, w.img, w.mask))),
                     ("forward_graph", lambda: w.model.forward_graph(w.ev, w.img_src, w.mask))):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        print(f"{cfg} B=1 {name}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per pair")
