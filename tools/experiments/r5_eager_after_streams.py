"""Which part of EventStage slows later EAGER single-pair forwards by ~0.3 ms (bench: 0.76 -> 1.06 ms after SameTimeEvaluator.run)?
    gpurun -- python tools/experiments/r5_eager_after_streams.py"""
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
wl = bench.Workload(pkg, dev, "sp_mnn", 1)


def t(tag):
    sec, _ = wl.timed(200, init=5)
    print(f"{tag:60s} {sec * 1e3:.3f} ms", flush=True)


t("baseline")
x = torch.empty(1 << 20, device=dev)
s2 = torch.cuda.Stream(dev)
t("after creating a second stream (unused)")
with torch.cuda.stream(s2):
    x.add_(1.0)
torch.cuda.synchronize()
t("after a kernel on the second stream")
h = torch.empty(1 << 20, dtype=torch.float32, pin_memory=True)
t("after a pinned allocation")
x.copy_(h, non_blocking=True)
torch.cuda.synchronize()
t("after a non-blocking H2D copy on the current stream")
with torch.cuda.stream(s2):
    x.copy_(h, non_blocking=True)
torch.cuda.synchronize()
t("after a non-blocking H2D copy on the second stream")
e = torch.cuda.Event()
e.record(s2)
torch.cuda.current_stream().wait_event(e)
torch.cuda.synchronize()
t("after event record on s2 + wait on the current stream")
for _ in range(3):
    with torch.cuda.stream(s2):
        x.copy_(h, non_blocking=True)
    e = torch.cuda.Event()
    e.record(s2)
    torch.cuda.current_stream().wait_event(e)
    x.mul_(1.0)
torch.cuda.synchronize()
t("after three copy -> event -> wait -> kernel rounds")
