"""Timing of einx_score_map (softmax over 65 channels + pixel shuffle + events mask / border) at the bench shapes.
   python tools/score_bench.py [B ...]   EINX_LIB=ab_libs/libeinx_X.so selects an A/B build."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("ei-nexus_official_amd")
nat = pkg.native


def timed(f, n=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for B in [int(a) for a in sys.argv[1:] if a.isdigit()] or [32, 1]:
    logits = torch.randn(B, 65, 33, 44, device="cuda")
    mask = torch.rand(B, 1, 260, 346, device="cuda") > 0.7
    for name, m in (("image side (no mask)", None), ("event side (mask, dilated)", mask)):
        us = timed(lambda: nat.score_map(logits, m, (3, 3, 2, 2), True, 4))
        print(f"score65 B={B} {name}: {us:.1f} us")
    l1 = torch.randn(B, 1, 260, 346, device="cuda")
    for name, m in (("no mask", None), ("mask, dilated", mask)):
        us = timed(lambda: nat.score_map(l1, m, (0, 0, 0, 0), True, 4))
        print(f"score1  B={B} {name}: {us:.1f} us")
