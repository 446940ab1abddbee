#!/usr/bin/env python3
"""Diagnostic: what does an initialised RCCL process group change for the step loop?  (B=32 SP+MNN; wall per
forward, device back-to-back time, host event-wait latency, CPU affinity before / after init)."""
import importlib
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(tag, model, ev, img0, img, mask):
    for _ in range(5):
        img.copy_(img0)
        model(ev, img, mask)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        img.copy_(img0)
        model(ev, img, mask)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        model.forward_batched(ev, img, mask)
    e1.record()
    torch.cuda.synchronize()
    devt = e0.elapsed_time(e1) / n
    # latency of waking up from an event wait: tiny kernel + event + synchronize
    x = torch.zeros(8, device=ev.device)
    lat = 0.0
    for _ in range(200):
        x.add_(1.0)
        e = torch.cuda.Event()
        e.record()
        t1 = time.perf_counter()
        e.synchronize()
        lat += time.perf_counter() - t1
    print(f"{tag:28s} wall {wall * 1e3:.3f} ms  device back-to-back {devt:.3f} ms  event-wait {lat / 200 * 1e6:.1f} us  "
          f"affinity {len(os.sched_getaffinity(0))} cpus", flush=True)


def main():
    pkg = importlib.import_module("ei-nexus_official_amd")
    synth = pkg.synth
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    cfg = pkg.default_config("SP_MNN", event_channels=5)
    model = pkg.EIM(cfg, device=dev).eval()
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=11)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    for ext in (model.event_extractor.extractor, model.image_extractor.extractor):
        ext.dense_outputs = False
    model.matcher.matcher.want_log_assignment = False
    B = 32
    ev, mask = synth.synth_events(10_000, B, 5)
    img0 = torch.from_numpy(synth.synth_image(10_000, B)).to(dev)
    ev, mask = torch.from_numpy(ev).to(dev), torch.from_numpy(mask).to(dev)
    img = img0.clone()
    measure("before init_process_group", model, ev, img0, img, mask)
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
    mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
    if mode == "eager":
        dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)
    else:
        dist.init_process_group(backend="nccl", init_method="env://")
    measure("after init (no collective)", model, ev, img0, img, mask)
    t = torch.ones(8, dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    measure("after first all_reduce", model, ev, img0, img, mask)
    dist.destroy_process_group()
    measure("after destroy_process_group", model, ev, img0, img, mask)


if __name__ == "__main__":
    main()
