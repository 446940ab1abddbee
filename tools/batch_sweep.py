"""SP+MNN forward time over batch sizes (synchronous EIM.forward, inputs resident): pairs/s and ms per forward, plus the share of the
fp32 MFMA peak the extraction FLOPs reach end to end.   python tools/batch_sweep.py [config]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("ei-nexus_official_amd")
dev = torch.device("cuda", 0)
cfg = sys.argv[1] if len(sys.argv) > 1 else "sp_mnn"
GF = {"sp_mnn": 31.95 + 0.54, "sp_lg": 31.95 + 80.5, "silk_mnn": 365.0}.get(cfg, 32.5)
for B in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64):
    w = bench.Workload(pkg, dev, cfg, B)
    step = lambda: (w.img.copy_(w.img_src), w.model(w.ev, w.img, w.mask))
    for _ in range(8):
        step()
    n = max(10, min(200, int(400 / B)))
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    print(f"B={B:3d}: {best * 1e3:8.3f} ms/forward  {B / best:8.1f} pairs/s  {best * 1e3 / B:7.3f} ms/pair  {GF * B / best / 1e3:6.1f} TFLOP/s = {GF * B / best / 1e3 / 157.3:.2f} of peak", flush=True)
    del w
    torch.cuda.empty_cache()
