#!/usr/bin/env python3
"""Single-pair latency of EIM.forward (the reference's own call pattern, test_events-image_same-time.py:130-194):
wall time per forward, host enqueue time (forward_batched returns before the device is done) and device time.
    python tools/latency_b1.py [B [SP_MNN|SP_LG|SiLK_MNN|SiLK_LG]]"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ei-nexus_official_amd")
synth = pkg.synth


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    name = sys.argv[2] if len(sys.argv) > 2 else "SP_MNN"  # SP_MNN | SP_LG | SiLK_MNN | SiLK_LG
    dev = "cuda:0"
    cfg = pkg.default_config(name, event_channels=5)
    model = pkg.EIM(cfg, device=dev).eval()
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=11)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    for ext in (model.event_extractor.extractor, model.image_extractor.extractor):
        ext.dense_outputs = False
    if name.endswith("MNN"):
        model.matcher.matcher.want_log_assignment = False
    ev, mask = synth.synth_events(10_000, B, 5)
    img0 = torch.from_numpy(synth.synth_image(10_000, B)).to(dev)
    ev, mask = torch.from_numpy(ev).to(dev), torch.from_numpy(mask).to(dev)
    img = img0.clone()
    for _ in range(20):
        img.copy_(img0)
        model(ev, img, mask)
    torch.cuda.synchronize()
    n = 200
    t_wall = t_enq = 0.0
    for _ in range(n):
        img.copy_(img0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model(ev, img, mask)
        t_wall += time.perf_counter() - t0
    for _ in range(n):
        img.copy_(img0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.forward_batched(ev, img, mask)
        t_enq += time.perf_counter() - t0
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        model.forward_batched(ev, img, mask)
    e1.record()
    torch.cuda.synchronize()
    dev_ms = e0.elapsed_time(e1) / n
    mode = "einx_extract (handle-level ABI)"
    print(f"B={B} {name} {mode}: forward wall {t_wall / n * 1e3:.3f} ms, host enqueue {t_enq / n * 1e3:.3f} ms, "
          f"back-to-back device+enqueue {dev_ms:.3f} ms per forward, overlap={model.overlap_extractors}")


if __name__ == "__main__":
    main()
