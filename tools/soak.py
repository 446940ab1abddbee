#!/usr/bin/env python3
"""Soak: many forwards (synchronous and streamed) -- throughput stays flat, allocator footprint stays flat, outputs stay identical."""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pkg = importlib.import_module("ei-nexus_official_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32  # 1: single pairs (conv16 + forked head branches)
DENSE = "--dense" in sys.argv
CFG = next((a for a in sys.argv[2:] if a in bench.WORKLOADS), "sp_mnn")  # e.g. sp_lg: stacked LightGlue sides, small-grid linears at B < 8
wl = bench.Workload(pkg, torch.device("cuda", 0), CFG, B, dense=DENSE, log_assignment=DENSE)
ref = wl.step()
ref_pos = [p.clone() for p in ref[0]["sparse_positions"]]
ref_m = [m.clone() for m in ref[2]["matches0"]]
ref_la = [m.clone() for m in ref[2]["matching_scores0"]] if CFG.endswith('lg') else []
for chunk in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = (150 if B >= 8 else 600) // (4 if CFG.endswith('lg') else 1)
    if chunk % 2 == 0:
        for _ in range(n):
            out = wl.step()
    else:
        def gen():
            for _ in range(n):
                wl.img.copy_(wl.img_src)
                yield (wl.ev, wl.img, wl.mask)
        for out in wl.model.forward_stream(gen()):
            pass
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = all(torch.equal(a, b) for a, b in zip(out[0]["sparse_positions"], ref_pos)) and all(torch.equal(a, b) for a, b in zip(out[2]["matches0"], ref_m)) and all(torch.equal(a, b) for a, b in zip(out[2]["matching_scores0"], ref_la))
    print(f"chunk {chunk} ({'sync' if chunk % 2 == 0 else 'stream'}): {B * n / dt:7.1f} pairs/s, allocated {torch.cuda.memory_allocated() / 1e6:8.1f} MB, "
          f"reserved {torch.cuda.memory_reserved() / 1e6:8.1f} MB, outputs identical: {same}", flush=True)
