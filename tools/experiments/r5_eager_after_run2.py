"""bench.py's own sequence: harness legs (incl. the streamed loop, which creates a copy stream), then a NEW single-pair workload:
per-forward times.  Before einx_fork_stream_prepare the new model's fork streams landed on their callers' pipes: 1.06 ms instead of 0.77."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

bench._import_shard_only("placement").cap_thread_pools()
import torch  # noqa: E402

pkg = importlib.import_module("ei-nexus_official_amd")
dev = torch.device("cuda", 0)
wl = bench.Workload(pkg, dev, "sp_mnn", 32)
wl.timed(5)
legs = bench.harness_leg(pkg, wl, torch)
print([l["ms_per_step"] for l in legs])
w = bench.Workload(pkg, dev, "sp_mnn", 1)
for _ in range(3):
    w.step()
torch.cuda.synchronize()
ts = []
for _ in range(60):
    t0 = time.perf_counter()
    w.step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("per forward ms:", " ".join(f"{v:.2f}" for v in ts))
print("mean", sum(ts) / len(ts), "median", sorted(ts)[len(ts) // 2])
