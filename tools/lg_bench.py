#!/usr/bin/env python3
"""Matcher-only micro-benchmarks on the GPU box (not part of bench.py's contract):
  * einx_linear at the LightGlue GEMM shapes (TFLOP/s per shape)
  * LightGlue.match_batched alone at B pairs x 1024 keypoints (ms, TFLOP/s)
  * MNN alone
Usage: python tools/lg_bench.py [--batch 64] [--reps 10]"""
import argparse
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ei-nexus_official_amd")
N = pkg.native


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--skip-linear", action="store_true")
    ap.add_argument("--only-linear", action="store_true")
    ap.add_argument("--data", default="randn", choices=["randn", "zeros", "relu"], help="operand values of the linear shapes (power / clock sensitivity)")
    ap.add_argument("--widths", action="store_true", help="LightGlue.match_batched at other (num_heads, head_dim) configurations and exit")
    a = ap.parse_args()
    dev = "cuda:0"
    B, n = a.batch, 1024
    if a.widths:
        from importlib import import_module
        bt = import_module("ei-nexus_official_amd.core.modules.matchers._batched")
        for heads, dh in ((4, 64), (8, 32), (2, 128), (4, 32), (3, 64), (4, 128)):
            d = heads * dh
            lg = pkg.LightGlue({"input_dim": d, "descriptor_dim": d, "num_heads": heads}).to(dev).eval()
            lg.want_log_assignment = False
            pbs = []
            for s_ in (1, 2):
                pb = bt.PairBatch()
                g = torch.Generator(device=dev).manual_seed(s_)
                pb.kpts = torch.rand(B, n, 3, device=dev, generator=g) * 250.0
                pb.desc = torch.nn.functional.normalize(torch.randn(B, n, d, device=dev, generator=g), dim=-1).contiguous()
                pb.counts = torch.full((B,), n, dtype=torch.int32, device=dev)
                pb.cap, pb.B, pb.image_size, pb.counts_host = n, B, (260, 346), [n] * B
                pbs.append(pb)
            ms = timed(lambda: lg.match_batched(pbs[0], pbs[1]), a.reps)
            # per pair and layer (out_proj / to_out folded): linears 2 n (3 d^2 + 4 d^2 + 2 d^2) + 2 n (2 d^2 + 4 d^2 + 2 d^2) MACs, attention 4 x 2 n^2 d
            flop = 9 * (2.0 * 2 * n * 17 * d * d + 2.0 * 4 * 2 * n * n * d) + 2.0 * 2 * n * d * d + 2.0 * n * n * d
            print(f"lightglue B={B} {heads} x {dh} (d={d}): {ms:8.2f} ms  {flop * B / ms / 1e9:7.1f} TFLOP/s", flush=True)
        return
    if not a.skip_linear:
        for (K, Nn) in ((256, 768), (256, 256), (512, 512), (512, 256)):
            M = B * n
            x = torch.randn(M, K, device=dev)
            w = torch.randn(Nn, K, device=dev)
            if a.data == "zeros":
                x.zero_()
                w.zero_()
            elif a.data == "relu":
                x.relu_()
            b = torch.randn(Nn, device=dev)
            y = torch.empty(M, Nn, device=dev)
            ms = timed(lambda: N.linear(x, w, b, out=y), a.reps)
            print(f"linear M={M} K={K} N={Nn}: {ms * 1e3:8.1f} us  {2.0 * M * K * Nn / ms / 1e9:7.1f} TFLOP/s", flush=True)
    if a.only_linear:
        return
    from importlib import import_module
    bt = import_module("ei-nexus_official_amd.core.modules.matchers._batched")
    lg = pkg.LightGlue({"input_dim": 256}).to(dev).eval()
    pbs = []
    for s in (1, 2):
        pb = bt.PairBatch()
        g = torch.Generator(device=dev).manual_seed(s)
        pb.kpts = torch.rand(B, n, 3, device=dev, generator=g) * 250.0
        pb.desc = torch.nn.functional.normalize(torch.randn(B, n, 256, device=dev, generator=g), dim=-1).contiguous()
        pb.counts = torch.full((B,), n, dtype=torch.int32, device=dev)
        pb.cap, pb.B, pb.image_size, pb.counts_host = n, B, (260, 346), [n] * B
        pbs.append(pb)
    lg.want_log_assignment = False
    ms = timed(lambda: lg.match_batched(pbs[0], pbs[1]), a.reps)
    print(f"lightglue B={B}: {ms:8.2f} ms  {80.5e9 * B / ms / 1e9:7.1f} TFLOP/s (80.5 GFLOP/pair nominal, folded projections do less)", flush=True)
    mnn = pkg.NearestNeighborMatcher()
    mnn.want_log_assignment = False
    ms = timed(lambda: mnn.match_batched(pbs[0], pbs[1]), a.reps)
    print(f"mnn B={B}: {ms * 1e3:8.1f} us  {0.537e9 * B / ms / 1e9:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
