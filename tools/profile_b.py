#!/usr/bin/env python3
"""per-kernel-class device time of one forward at batch B (library-side HIP-event scopes, single stream)
    python tools/profile_b.py [B] [config]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pkg = importlib.import_module("ei-nexus_official_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = sys.argv[2] if len(sys.argv) > 2 else "sp_mnn"
wl = bench.Workload(pkg, torch.device("cuda", 0), cfg, B)
wl.model.overlap_extractors = False
for _ in range(5):
    wl.step()
torch.cuda.synchronize()
prof = bench.library_profile(pkg, lambda: (wl.step(), torch.cuda.synchronize()))
tot = 0.0
for k, (c, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:55s} {c:4d} launches {ms * 1e3:9.1f} us")
    tot += ms
print(f"sum of profiled kernels {tot * 1e3:.1f} us at B={B}")
bench.layer_table(wl)
