"""Scratch (round 6): where a batch of SameTimeEvaluator.run goes -- host phases (pack, enqueue, finish) against the device's step."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("ei-nexus_official_amd")
rep = importlib.import_module("ei-nexus_official_amd.datasets.representations")
dev = torch.device("cuda", 0)
B = 32
w = bench.Workload(pkg, dev, "sp_mnn", B)
ev = pkg.SameTimeEvaluator(w.model, w.ce, (346, 260))
events = [pkg.synth.synth_raw_events(5000 + b, 60000) for b in range(B)]


def feed(n):
    for _ in range(n):
        yield events, w.img_src.clone()


for _ in ev.run(feed(4)):
    pass
torch.cuda.synchronize()
# forward only, device resident
for _ in range(3):
    w.img.copy_(w.img_src); w.model(w.ev, w.img, w.mask)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    w.img.copy_(w.img_src); w.model(w.ev, w.img, w.mask)
torch.cuda.synchronize()
print("forward only: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
for depth in (2, 3, 2, 3):
    ts = []
    for rep_i in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in ev.run(feed(40), depth=depth):
            pass
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 40 * 1e3)
    print("run depth %d: %s ms per batch" % (depth, " ".join("%.3f" % t for t in ts)))
ts = []
for rep_i in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        ev.step(events, w.img_src.clone())
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 30 * 1e3)
print("step: %s ms per batch" % " ".join("%.3f" % t for t in ts))
ts = []
for rep_i in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        w.img.copy_(w.img_src); w.model(w.ev, w.img, w.mask)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 30 * 1e3)
print("forward only again: %s ms" % " ".join("%.3f" % t for t in ts))
# phases of the loop body, by hand (depth 2)
from collections import deque
stage = [rep.EventStage(dev), rep.EventStage(dev)]
acc = {"pack": 0.0, "rep_enqueue": 0.0, "model_enqueue": 0.0, "finish": 0.0, "account": 0.0, "clone": 0.0}
pending = deque()
n = 20
torch.cuda.synchronize()
T0 = time.perf_counter()
for k in range(n):
    t0 = time.perf_counter()
    img = w.img_src.clone()
    t1 = time.perf_counter()
    packed = stage[k % 2].pack(events)
    t2 = time.perf_counter()
    grid = rep.events_to_voxel_grid_batch(events, (w.ce, 260, 346), True, dev, packed=packed)
    mask = rep.events_mask_batch(events, (346, 260), dev, packed=packed)
    t3 = time.perf_counter()
    p = w.model._enqueue(grid, img, mask, slot=k % 2)
    t4 = time.perf_counter()
    pending.append(p)
    t5 = t6 = t4
    if len(pending) >= 2:
        r = w.model._finish(pending.popleft())
        t5 = time.perf_counter()
        ev._account(*r, None)
        t6 = time.perf_counter()
    for key, a, b in (("clone", t0, t1), ("pack", t1, t2), ("rep_enqueue", t2, t3), ("model_enqueue", t3, t4), ("finish", t4, t5), ("account", t5, t6)):
        acc[key] += (b - a) * 1e3
while pending:
    ev._account(*w.model._finish(pending.popleft()), None)
torch.cuda.synchronize()
print("by hand: %.3f ms per batch; host phases (ms per batch):" % ((time.perf_counter() - T0) / n * 1e3), {k: round(v / n, 3) for k, v in acc.items()})
# device-only: representation + forward + metrics with everything resident
x, y, t, p_, offs = rep._pack(events, dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    grid = rep.events_to_voxel_grid_batch(events, (w.ce, 260, 346), True, dev, packed=(x, y, t, p_, offs))
    mask = rep.events_mask_batch(events, (346, 260), dev, packed=(x, y, t, p_, offs))
    w.img.copy_(w.img_src)
    r = w.model(grid, w.img, mask)
    ev._account(*r, None)
torch.cuda.synchronize()
print("resident events -> rep + forward + metrics: %.3f ms per batch" % ((time.perf_counter() - t0) / 20 * 1e3))
print("pack threads", rep.EventStage.pack_threads())
