#!/usr/bin/env python3
"""bench.py -- event-image pairs/s (extract + match, 346x260, 1024 keypoints) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config sp_mnn|silk_mnn|sp_lg] [--batch B]

A step = one pass of the hot path (EIM.forward: event extractor + image extractor + matcher) over
one batch of synthetic pairs already resident in HBM.  Default workload = BASELINE.json configs[1]:
batch 32, 5-bin event voxel + gray image, SuperPoint-shaped extractors + MNN matcher.
One process per GPU (torchrun sets RANK/LOCAL_RANK/WORLD_SIZE); pairs are independent, so ranks
shard them with NO data-path collective; the only RCCL traffic is the all-reduce of the metric
accumulators (weak scaling: the per-GPU batch is fixed).
Prints ONE JSON line (rank 0) with the driver contract plus `roofline` and `cpu_baseline`.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X dense fp32 matrix peak (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    "sp_mnn": ("SP_MNN", "B32 346x260 5-bin event voxel + gray image, VGG(event)+SuperPoint(image) extractors, MNN matcher, k=1024"),
    "silk_mnn": ("SiLK_MNN", "B32 346x260, VGG_NP(event)+SiLK(image) extractors, MNN matcher, k=1024"),
    "sp_lg": ("SP_LG", "B64 346x260, VGG(event)+SuperPoint(image) extractors, LightGlue matcher, k=1024"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="sp_mnn", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="pairs per GPU per step (default 32; 64 for sp_lg)")
    ap.add_argument("--log-assignment", action="store_true", help="also materialise log_assignment (reference-complete matcher dict)")
    ap.add_argument("--dense", action="store_true", help="also materialise the dense descriptor maps (reference-complete dict)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-torch", action="store_true", help="also time oracle/torch_cpu.py (plain PyTorch on the host cores; SP+MNN only)")
    ap.add_argument("--cpu-pairs", type=int, default=None, help="pairs of the CPU baseline sample (default: ~10-20 s of host work)")
    ap.add_argument("--with-metrics", action="store_true", help="also compute MR/MMA/VDD on the device each step (metrics.hip) and all-reduce their sums")
    ap.add_argument("--layer-table", action="store_true", help="tuning aid: time every conv layer of both extractors standalone and exit")
    ap.add_argument("--kernel-only", action="store_true", help="only run the dominant-kernel loop (for rocprofv3 --pmc passes)")
    return ap.parse_args()


def conv_layer_flops(cin, cout, ks, H, W):
    return 2.0 * cin * cout * ks * ks * H * W


def sp_pair_flops(ce):
    """algorithmic FLOPs of the two SuperPoint-shaped encoders + heads for one pair."""
    def net(c0):
        f = conv_layer_flops(c0, 64, 3, 264, 352) + conv_layer_flops(64, 64, 3, 264, 352)
        f += 2 * conv_layer_flops(64, 64, 3, 132, 176)
        f += conv_layer_flops(64, 128, 3, 66, 88) + conv_layer_flops(128, 128, 3, 66, 88)
        f += 2 * conv_layer_flops(128, 128, 3, 33, 44)
        f += 2 * conv_layer_flops(128, 256, 3, 33, 44) + conv_layer_flops(256, 65, 1, 33, 44) + conv_layer_flops(256, 256, 1, 33, 44)
        return f
    return net(ce) + net(1)


def main():
    args = parse()
    rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    distributed = "RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ
    if distributed:  # torchrun launch (also with one rank): RCCL process group, one process per GPU
        dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    pkg = importlib.import_module("ei-nexus_official_amd")
    synth = pkg.synth
    cfg_name, wl_desc = WORKLOADS[args.config]
    B = args.batch or (64 if args.config == "sp_lg" else 32)
    ce = 5
    cfg = pkg.default_config(cfg_name, event_channels=ce)
    model = pkg.EIM(cfg, device=dev).eval()
    sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=11)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    for ext in (model.event_extractor.extractor, model.image_extractor.extractor):
        ext.dense_outputs = bool(args.dense)
    model.matcher.matcher.want_log_assignment = bool(args.log_assignment)

    # synthetic pairs: each rank gets its own shard of the global pair index space
    ev_np, mask_np = synth.synth_events(10_000 + rank * B, B, ce)
    img_np = synth.synth_image(10_000 + rank * B, B)
    ev = torch.from_numpy(ev_np).to(dev)
    mask = torch.from_numpy(mask_np).to(dev)
    img_src = torch.from_numpy(img_np).to(dev)
    img = torch.empty_like(img_src)

    acc = pkg.shard.MetricAccumulator(dev)  # pairs, keypoints(ev), keypoints(im), matches, ...

    metric_sums = torch.zeros(9, dtype=torch.float64, device=dev)
    metric_rows = []
    batch_metrics = importlib.import_module(pkg.__name__ + ".core.metrics._native_metrics").batch_metrics

    def step(accumulate=False):
        img.copy_(img_src)  # SuperPoint scales its input in place (reference quirk), so refresh it
        ef, imf, m = model(ev, img, mask)
        res = None
        if args.with_metrics:  # harness metrics of the reference's test script, computed on the device
            res = batch_metrics(ef._batched, imf._batched, model._last_match)
        if accumulate:
            acc.add_batch(ef, imf, m)
            if res is not None:
                metric_rows.append(res)  # [B,9] per step; summed after the timed region
        return ef, imf, m

    if args.layer_table:
        rows = []
        for _ in range(3):  # bring the device to its working clocks before the first timed row
            step()
        torch.cuda.synchronize()
        for side, ext, x0 in (("event", model.event_extractor.extractor, ev), ("image", model.image_extractor.extractor, img_src)):
            eng = ext.engine()
            pads = pkg.native.padder_pads(260, 346, ext.cell_size)
            Hp, Wp = 260 + pads[2] + pads[3], 346 + pads[0] + pads[1]
            chains = [("bb", eng.backbone), ("det", eng.det_head), ("desc", eng.desc_head)]
            feats = None
            for cname, layers in chains:
                t_in = x0 if cname == "bb" else feats
                for li, layer in enumerate(layers):
                    fold = (pads[2], pads[0], Hp, Wp) if (cname == "bb" and li == 0) else None
                    out = layer(t_in, fold=fold)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        layer(t_in, fold=fold)
                    e1.record()
                    torch.cuda.synchronize()
                    dur = e0.elapsed_time(e1) / 5 * 1e-3
                    H_, W_ = (Hp, Wp) if fold else t_in.shape[-2:]
                    fl = conv_layer_flops(layer.cin, layer.cout, layer.ks, H_, W_) * B
                    rows.append((f"{side}.{cname}{li}", layer.cin, layer.cout, layer.ks, int(H_), int(W_), bool(layer.pool), round(dur * 1e6, 1),
                                 round(fl / dur / 1e12, 1)))
                    t_in = out
                if cname == "bb":
                    feats = t_in
        for r in rows:
            print("%-14s cin=%3d cout=%3d ks=%d %3dx%3d pool=%d  %8.1f us  %6.1f TFLOP/s" % r)
        print("total conv us", round(sum(r[7] for r in rows), 1))
        return
    if args.kernel_only:
        args.steps, args.warmup, args.no_cpu_baseline = 0, 0, True
    else:
        # initialisation, not a measured or warm-up step: the first forward builds the kernel-native weight images
        # (repack, BN fold, LightGlue projection folding), loads the code objects, sizes the allocator pool and settles
        # the sticky NMS pass budget; do it here so that `--warmup 0` does not time a cold start
        for _ in range(2):
            step()
    for _ in range(args.warmup):
        step()

    def barrier():
        if distributed:
            dist.barrier(device_ids=[local])

    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(accumulate=True)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    acc.all_reduce()  # the one collective of the job: metric accumulators (RCCL over xGMI when world > 1)
    if args.with_metrics and metric_rows:
        metric_sums = torch.nan_to_num(torch.cat(metric_rows)).sum(0)
    if args.with_metrics and distributed:
        dist.all_reduce(metric_sums, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    stats = acc.as_dict()
    pairs_total = max(stats["pairs"], 1.0)
    value = stats["pairs"] / elapsed

    # ---- roofline of the dominant kernel: the second backbone conv (64->64 at full resolution) ----------
    # SP-shaped nets: conv_block_kernel<3,8,32,2,4,1,2,8,true,true> (conv1b, fused pool, offset-table reloads); SiLK: same tile, no pool.
    roofline = None
    if rank == 0:
        ext = model.image_extractor.extractor
        eng = ext.engine()
        l0, l1 = eng.backbone[0], eng.backbone[1]
        pads = pkg.native.padder_pads(260, 346, ext.cell_size)
        Hp, Wp = 260 + pads[2] + pads[3], 346 + pads[0] + pads[1]
        x1 = l0(img_src, fold=(pads[2], pads[0], Hp, Wp))
        reps = 10
        if args.kernel_only:
            # no pipeline steps ran before: bring the device to its working clocks with the *first* layer's kernel, so that
            # every launch of the measured kernel in a `rocprofv3 --stats` summary of this mode is a steady-state launch
            for _ in range(300):
                l0(img_src, fold=(pads[2], pads[0], Hp, Wp))
        for _ in range(2):
            l1(x1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()  # HIP events on the stream the kernel is launched on (torch's current stream)
        for _ in range(reps):
            l1(x1)
        e1.record()
        torch.cuda.synchronize()
        dur = e0.elapsed_time(e1) * 1e-3 / reps
        flops = conv_layer_flops(l1.cin, l1.cout, l1.ks, Hp, Wp) * B
        ach = flops / dur / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_conv1b.json")
        if args.config == "sp_mnn" and B == 32 and os.path.exists(pmc):
            traffic = json.load(open(pmc))["hbm_bytes_per_launch"]  # FETCH_SIZE+WRITE_SIZE, separate --pmc passes
        kname = f"conv_block_kernel<3,8,32,2,4,1,2,8,{'true' if l1.pool else 'false'},true> ({l1.cin}->{l1.cout} 3x3 @{Hp}x{Wp}, B={B})"
        roofline = {"kernel": kname, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic, "launch_ms": round(dur * 1e3, 4),
                    "flop_per_launch": flops,
                    "hbm_frac_at_measured_rate": round(188.8e6 * value / max(world, 1) / 8e12, 4) if args.config == "sp_mnn" else None}

    # ---- CPU baseline: the oracle (a port, not the reference files) on the host cores, bounded sample ----
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # reported at N=1 only
        from oracle import oracle as orc
        nb = args.cpu_pairs if args.cpu_pairs else {"sp_mnn": 32, "sp_lg": 4, "silk_mnn": 4}.get(args.config, 4)
        nb = max(1, min(nb, B))
        sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}  # noqa: E731
        et, it = cfg.event_extractor.type, cfg.image_extractor.type
        escale, iscale = cfg.event_extractor[et].descriptor_scale_factor, cfg.image_extractor[it].descriptor_scale_factor
        passes = 2 if (args.config == "sp_mnn" and not args.cpu_pairs) else 1  # the headline sample: ~10-20 s of host work
        tc = time.perf_counter()
        nmatch = 0
        for _ in range(passes):
            oe = orc.extractor_forward(et, sub("event_extractor.extractor."), ev_np[:nb].copy(), mask_np[:nb], top_k=1024, scale=escale)
            oi = orc.extractor_forward(it, sub("image_extractor.extractor."), img_np[:nb].copy(), None, top_k=1024, scale=iscale)
            for b in range(nb):
                if cfg.matcher.type == "MNN":
                    r = orc.mnn(oe["sparse_descriptors"][b], oi["sparse_descriptors"][b], want_la=args.log_assignment)
                else:
                    r = orc.lightglue(sub("matcher.matcher."), oe["sparse_positions"][b], oe["sparse_descriptors"][b],
                                      oi["sparse_positions"][b], oi["sparse_descriptors"][b])
                nmatch += int((r["matches0"] > -1).sum())
        cpu_s = time.perf_counter() - tc
        nb *= passes
        cores = os.cpu_count() or 1
        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:
            pass
        cpu_baseline = {"value": round(nb / cpu_s, 3), "unit": "pairs/s", "cores": cores, "kind": "port",
                        "sample": f"{nb} pairs of the same workload through oracle/ (C, OpenMP on all host cores), {cpu_s:.1f} s"}

    cpu_torch = None
    if rank == 0 and world == 1 and args.cpu_torch and args.config == "sp_mnn":
        from oracle import torch_cpu
        sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}  # noqa: E731
        nb = min(16, B)
        torch_cpu.sp_mnn_pairs(sub("event_extractor.extractor."), sub("image_extractor.extractor."), ev_np[:2], mask_np[:2], img_np[:2].copy())
        tc = time.perf_counter()
        torch_cpu.sp_mnn_pairs(sub("event_extractor.extractor."), sub("image_extractor.extractor."), ev_np[:nb], mask_np[:nb], img_np[:nb].copy())
        cpu_s = time.perf_counter() - tc
        cpu_torch = {"value": round(nb / cpu_s, 3), "unit": "pairs/s", "threads": torch.get_num_threads(), "kind": "plain PyTorch CPU expression (oracle/torch_cpu.py)",
                     "sample": f"{nb} pairs, one batched call, {cpu_s:.1f} s"}

    if rank == 0:
        out = {
            "metric": "event-image pairs/s (extract+match, 346x260, 1024 kpts)",
            "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl_desc, "pairs_per_gpu_per_step": B, "global_batch": B * world, "event_bins": ce,
                       "parallelism": f"dp{world} (pairs sharded, metric all-reduce only)", "log_assignment": bool(args.log_assignment),
                       "dense_outputs": bool(args.dense),
                       "mean_keypoints": [round(stats["keypoints0"] / pairs_total, 1), round(stats["keypoints1"] / pairs_total, 1)],
                       "mean_matches": round(stats["matches"] / pairs_total, 1),
                       "harness_metrics_mean": ([round(v, 5) for v in (metric_sums / pairs_total).tolist()] if args.with_metrics else None)},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        if cpu_torch is not None:
            out["cpu_baseline_torch"] = cpu_torch
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
