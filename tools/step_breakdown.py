#!/usr/bin/env python3
"""Host-side breakdown of one EIM.forward at the bench shape (tuning aid): how long the host spends
enqueueing, waiting for the device, and building the output dicts."""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ei-nexus_official_amd")
synth = pkg.synth


def main():
    dev = "cuda:0"
    B = int(os.environ.get("B", 32))
    cfg = pkg.default_config("SP_MNN", event_channels=5)
    model = pkg.EIM(cfg, device=dev).eval()
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=7)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    for ext in (model.event_extractor.extractor, model.image_extractor.extractor):
        ext.dense_outputs = False
    model.matcher.matcher.want_log_assignment = False
    ev, mask = synth.synth_events(1234, B, 5, 260, 346)
    img0 = torch.from_numpy(synth.synth_image(1234, B, 260, 346)).to(dev)
    ev, mask = torch.from_numpy(ev).to(dev), torch.from_numpy(mask).to(dev)
    img = img0.clone()
    for _ in range(5):
        img.copy_(img0)
        model(ev, img, mask)
    torch.cuda.synchronize()
    rows = []
    for _ in range(20):
        t0 = time.perf_counter()
        img.copy_(img0)
        e, i, mr = model.forward_batched(ev, img, mask)
        t1 = time.perf_counter()
        e.prepare()
        i.prepare()
        pre = model.matcher.__class__.__module__ and __import__("importlib").import_module(pkg.__name__ + ".core.modules.matchers._batched").full_batch_lists(mr)
        t2 = time.perf_counter()
        host = torch.stack([e.det.counts, i.det.counts, e.det.not_converged, i.det.not_converged, mr.nmatch]).cpu()
        t3 = time.perf_counter()
        n, m = host[0].tolist(), host[1].tolist()
        ef = e.materialize(n)
        imf = i.materialize(m)
        t4 = time.perf_counter()
        mt = model.matcher.materialize(mr, n, m, host[4].tolist(), prebuilt=pre)
        t5 = time.perf_counter()
        rows.append([t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0])
    r = np.array(rows[5:]) * 1e3
    names = ["enqueue", "prepare", "wait(sync)", "feats lists", "match lists", "total"]
    for k, v in zip(names, r.mean(0)):
        print(f"{k:12s} {v:8.3f} ms")
    # same loop through the public forward
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        img.copy_(img0)
        model(ev, img, mask)
    torch.cuda.synchronize()
    print(f"forward      {(time.perf_counter() - t0) / 20 * 1e3:8.3f} ms/step")


if __name__ == "__main__":
    main()
