"""First GPU process on a fresh box: ms per step over time (is there a cold phase, and how long?).  argv: config batch seconds"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import torch
cfg, B, secs = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
pkg = importlib.import_module("ei-nexus_official_amd")
t_start = time.time()
wl = bench.Workload(pkg, torch.device("cuda", 0), cfg, B, rank=0, calibrate=True, dense=False, log_assignment=False)
print(f"{cfg} B={B}: model ready after {time.time() - t_start:.1f} s", flush=True)
t0 = time.time()
k = 0
while time.time() - t0 < secs:
    n = 5 if cfg.startswith("silk") else 40
    torch.cuda.synchronize()
    a = time.perf_counter()
    for _ in range(n):
        wl.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - a) / n * 1e3
    k += n
    print(f"t={time.time() - t0:6.2f} s  steps {k:5d}  {ms:8.3f} ms/step", flush=True)
