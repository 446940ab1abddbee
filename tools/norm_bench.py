"""Timing of einx_normalize_map (coarse descriptor map -> L2-normalised map + channels-last copy) at the bench shapes.
   python tools/norm_bench.py [B]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("ei-nexus_official_amd")
nat = pkg.native
Bs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [32, 1]
for B, D, P in [(b, 256, 33 * 44) for b in Bs] + [(Bs[0], 128, 33 * 44)]:
    raw = torch.randn(B, D, 33, 44, device="cuda")
    for _ in range(5):
        nat.normalize_map(raw, 1.0, want_cl=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        nat.normalize_map(raw, 1.0, want_cl=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    gb = 3 * raw.numel() * 4 / 1e9
    print(f"normalize_map B={B} D={D} P={P}: {us:.1f} us  {gb / us * 1e6 / 1e3:.2f} TB/s (read + 2 writes)")
