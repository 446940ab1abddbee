#!/usr/bin/env python3
"""bench.py -- event-image pairs/s (extract + match, 346x260, 1024 keypoints) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config sp_mnn|silk_mnn|sp_lg] [--batch B]

A step = one pass of the hot path (EIM.forward: event extractor + image extractor + matcher) over
one batch of synthetic pairs already resident in HBM.  Default workload = BASELINE.json configs[1]:
batch 32, 5-bin event voxel + gray image, SuperPoint-shaped extractors + MNN matcher.

One process per GPU.  Under a launcher (torchrun sets RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*) this
process is a rank; started plainly with `--gpus N` (N > 1) it is the LAUNCHER: it never touches
the GPU, spawns N fresh rank processes with that environment (the reference's pattern:
init_process_group(env://) under a launcher, train_extractor.py:82-91) and exits with their status.
Pairs are independent, so ranks shard them with NO data-path collective; the only RCCL traffic is
the all-reduce of the metric accumulators (weak scaling: the per-GPU batch is fixed).

Rank 0 prints ONE JSON line with the driver contract plus
  roofline         dominant kernel (conv1b), mean of >= 20 launches timed with HIP events
  roofline_stages  conv stage, descriptor-correlation GEMM, LightGlue GEMM / attention kernels
  cpu_baseline     the oracle (a port) on the host cores, bounded sample; `verified_pairs` = pairs of
                   that sample whose GPU outputs (keypoints, descriptors, matches) equal the oracle's
  scale_legs       all-rank legs run after the headline at EVERY N: BASELINE configs[4] = SP+LightGlue, 64 pairs per GPU
                   (whole-job pairs/s over the slowest rank, per-rank min/max, RCCL world)
  extra_configs    legs for BASELINE configs[2], configs[3], the reference-complete dict, B=1 latency and the round-1 weights
  cpu_baseline_torch  plain PyTorch on the host cores (oracle/torch_cpu.py), bounded to a few seconds
  rccl             world size seen by the process group + latency of the metric all-reduce (N > 1 or --spawn)
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# HIP maps streams onto a small pool of hardware queues (4 by default).  RCCL's own streams take some of them, and
# the event extractor's side stream then shares a queue with the main stream: the two extractors serialise and the
# step is 5 % slower (measured: 3380 -> 3200 pairs/s, back to 3375 with 8 queues; profiles/README.md).  The HIP
# runtime reads this when it initialises, which in this process happens after this line (torch is imported later).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X dense fp32 matrix peak (/opt/skills/guides/MI355X_MICROARCH.md)
PEAK_HBM_BYTES = 8.0e12
SP_PAIR_BYTES = 188.8e6  # algorithmic HBM bytes of the conv stage per pair
LG_PAIR_FLOP = 80.5e9    # LightGlue forward per pair at 1024 x 1024 keypoints

WORKLOADS = {
    "sp_mnn": ("SP_MNN", 32, "346x260 5-bin event voxel + gray image, VGG(event)+SuperPoint(image) extractors, MNN matcher, k=1024"),
    "silk_mnn": ("SiLK_MNN", 32, "346x260, VGG_NP(event)+SiLK(image) extractors, MNN matcher, k=1024"),
    "sp_lg": ("SP_LG", 64, "346x260, VGG(event)+SuperPoint(image) extractors, LightGlue matcher, k=1024"),
    "silk_lg": ("SiLK_LG", 32, "346x260, VGG_NP(event)+SiLK(image) extractors, LightGlue matcher (128-d input_proj), k=1024"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="sp_mnn", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="pairs per GPU per step (default 32; 64 for sp_lg)")
    ap.add_argument("--log-assignment", action="store_true", help="also materialise log_assignment (reference-complete matcher dict)")
    ap.add_argument("--dense", action="store_true", help="also materialise the dense descriptor maps (reference-complete dict)")
    ap.add_argument("--raw-weights", action="store_true",
                    help="headline on the un-calibrated synthetic weights (round-1 workload: near-constant descriptors, ~2 matches per pair)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip extra_configs and roofline_stages (they run at N=1 only)")
    ap.add_argument("--extras", action="store_true", help="run extra_configs / roofline_stages on rank 0 even when N > 1")
    ap.add_argument("--no-cpu-torch", action="store_true", help="skip the plain-PyTorch CPU leg (oracle/torch_cpu.py, ~5 s; SP+MNN at N=1 only)")
    ap.add_argument("--cpu-torch", action="store_true", help="(default now; kept for old command lines)")
    ap.add_argument("--cpu-pairs", type=int, default=None, help="pairs of the CPU baseline sample (default: ~10-20 s of host work)")
    ap.add_argument("--with-metrics", action="store_true", help="also compute MR/MMA/VDD on the device each step (metrics.hip) and all-reduce their sums")
    ap.add_argument("--layer-table", action="store_true", help="tuning aid: time every conv layer of both extractors standalone and exit")
    ap.add_argument("--kernel-only", action="store_true", help="only run the dominant-kernel loop (for rocprofv3 --pmc passes)")
    ap.add_argument("--no-scale-legs", action="store_true",
                    help="skip the all-rank BASELINE configs[4] leg (SP+LightGlue, 64 pairs per GPU) that follows the headline at every N")
    ap.add_argument("--spawn", action="store_true", help="go through the rank launcher even for --gpus 1 (one-rank RCCL group)")
    ap.add_argument("--dry-run-gloo", action="store_true",
                    help="launcher / collective rehearsal on CPU: ranks form a gloo group, all-reduce the metric accumulators, run no kernels")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------ launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """Parent of `--gpus N`: spawn N fresh rank processes (env:// rendezvous on 127.0.0.1) and wait.
    This process makes no GPU call.  A rank that dies takes the others down (exact PIDs), so a
    failure is loud and never a hang in the rendezvous."""
    n = args.gpus
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    live = set(range(n))
    while live:
        for r in list(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"[bench launcher] rank {r} exited with status {code}; stopping the other ranks", file=sys.stderr)
                for o in live:
                    procs[o].terminate()
        if live:
            time.sleep(0.05)
    for p in procs:
        if p.poll() is None:
            p.kill()
    return rc


def stream_overlap_report(pkg, dev):
    """The four streams of a forward (caller, the event extractor's side stream, the library's fork stream of each) probed pair by
    pair with einx_stream_overlap_us AFTER the timed region: elapsed / spin, 1 = side by side, 2 = serialised on one hardware queue.
    (What the step rate depends on besides the kernels: under a process group RCCL's streams take hardware queues as well.)"""
    import ctypes
    import torch
    N = importlib.import_module(pkg.__name__ + "._native")
    EIM = importlib.import_module(pkg.__name__ + ".core.modules.EIM").EIM
    lib = N.lib()
    cur = torch.cuda.current_stream(dev).cuda_stream
    side = EIM._side_streams.get((dev.type, dev.index, cur))
    if side is None:
        return None
    hs = {"main": cur, "side": side.cuda_stream, "fork(main)": lib.einx_fork_stream_of(ctypes.c_void_p(cur)),
          "fork(side)": lib.einx_fork_stream_of(ctypes.c_void_p(side.cuda_stream))}
    names = [k for k, v in hs.items() if k == "main" or v]
    out = {}
    spin = 200
    for i, a in enumerate(names):
        for b in names[i + 1:]:
            r = ctypes.c_float()
            if lib.einx_stream_overlap_us(ctypes.c_void_p(hs[a]), ctypes.c_void_p(hs[b]), spin, ctypes.byref(r)) == 0:
                out[f"{a}|{b}"] = round(r.value / spin, 2)
    return out


INIT_SECONDS = 3.0  # untimed forwards after the first two (see run_rank)


# ------------------------------------------------------------------------------------ all-rank legs
SCALE_LEG_STEPS = 10


def scale_leg_plan(args):
    """Workloads every rank runs after the headline, at every N (so that the N = 1, 2, 4, 8 lines of a scaling run each carry
    them): BASELINE configs[4] = SP + LightGlue, 64 pairs per GPU (512 over 8 GPUs), rank-sharded like the headline."""
    if args.no_scale_legs or args.kernel_only or args.layer_table or args.config == "sp_lg":
        return []
    return [("sp_lg", WORKLOADS["sp_lg"][1], SCALE_LEG_STEPS)]


def leg_record(config, batch, steps, world, elapsed_max, pairs_total, per_rank_pairs_s, mean_matches=None, calibrated=None):
    """one entry of `scale_legs` (rank 0): whole-job pairs/s over the slowest rank's time, plus the per-rank spread"""
    e = {"config": config, "workload": f"B{batch} " + WORKLOADS[config][2], "pairs_per_gpu_per_step": batch, "global_batch": batch * world,
         "n_gpus": world, "steps": steps, "value": round(pairs_total / elapsed_max, 2) if elapsed_max > 0 else 0.0, "unit": "pairs/s",
         "ms_per_step": round(elapsed_max / max(steps, 1) * 1e3, 3),
         "per_rank_pairs_per_s": {"min": round(min(per_rank_pairs_s), 2), "max": round(max(per_rank_pairs_s), 2)},
         "timing": "barrier + synchronize on both sides, max over ranks (as the headline)", "scaling": "weak"}
    if config == "sp_lg":
        e["note"] = "BASELINE configs[4] (Batch=512 SP+LightGlue over 8 GPUs = 64 per GPU); at N=1 this is configs[3]"
    if mean_matches is not None:
        e["mean_matches"] = round(mean_matches, 1)
    if calibrated is not None:
        e["calibrated_descriptors"] = bool(calibrated)
        e["same_scene_pairs"] = bool(calibrated) and config.endswith("_lg")
    return e


def all_rank_leg(pkg, dev, config, batch, steps, rank, world, dist=None, local=0, warmup=8):
    """Every rank builds the workload on its own shard of the pair index space and times `steps` forwards; returns
    (record on rank 0 | None, workload).  No data-path collective: a barrier on both sides, MAX of the elapsed time, SUM of pairs."""
    import torch
    w = Workload(pkg, dev, config, batch, rank=rank)
    for _ in range(2 + warmup):
        w.step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier(device_ids=[local])
    t0 = time.perf_counter()
    nm = 0
    for _ in range(steps):
        _, _, m = w.step()
        nm += sum(int(t.shape[0]) for t in m["matched_kpts0"])
    torch.cuda.synchronize()
    mine = time.perf_counter() - t0
    if dist is not None:
        dist.barrier(device_ids=[local])
    elapsed = time.perf_counter() - t0
    v = torch.tensor([elapsed, mine, float(steps * batch), float(nm)], dtype=torch.float64, device=dev)
    per_rank = [steps * batch / mine]
    if dist is not None:
        tmax = v[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        gathered = [torch.zeros_like(v) for _ in range(world)]
        dist.all_gather(gathered, v)
        elapsed = float(tmax.item())
        per_rank = [float(g[2].item() / g[1].item()) for g in gathered]
        pairs = sum(float(g[2].item()) for g in gathered)
        nm = sum(float(g[3].item()) for g in gathered)
    else:
        pairs = float(steps * batch)
    rec = leg_record(config, batch, steps, world, elapsed, pairs, per_rank, mean_matches=nm / max(pairs, 1.0), calibrated=w.calibrated) if rank == 0 else None
    return rec, w


# ------------------------------------------------------------------------------------ CPU rehearsal
def dry_run_gloo(args):
    """The N>1 control flow without kernels: same env:// rendezvous, same accumulator all-reduce,
    same barrier + max-over-ranks timing, on the gloo backend (tests/test_bench_launcher.py)."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    placement = place_this_rank(rank, int(os.environ.get("LOCAL_RANK", rank)), world)  # before torch is imported
    import torch
    import torch.distributed as dist
    finish_placement(placement, torch)
    if os.environ.get("EINX_BENCH_DRYRUN_FAIL_RANK") == str(rank):  # tests: one rank dies before the rendezvous completes
        raise SystemExit(7)
    dist.init_process_group(backend="gloo", init_method="env://")
    assert dist.get_world_size() == world, "process group does not span the launched ranks"
    placements = gather_placements(placement, dist, world)
    pkg_shard = _import_shard_only()
    B = args.batch or WORKLOADS[args.config][1]
    lo, hi = pkg_shard.shard_range(B * world, rank, world)  # the rank's block of the global pair index space
    ranges = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(ranges, torch.tensor([lo, hi], dtype=torch.int64))
    acc = pkg_shard.MetricAccumulator("cpu")
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        acc.add(B, B * 1000 + rank, B * 1001, 10 * (rank + 1), 0.0)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    acc.all_reduce()
    stats = acc.as_dict()
    legs = []
    for cfg_, b_, steps_ in scale_leg_plan(args):  # the same all-rank control flow as all_rank_leg, without kernels
        dist.barrier()
        t1 = time.perf_counter()
        time.sleep(0.001 * (rank + 1))
        mine = time.perf_counter() - t1
        dist.barrier()
        v = torch.tensor([time.perf_counter() - t1, mine, float(steps_ * b_), 0.0], dtype=torch.float64)
        tmax = v[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        gathered = [torch.zeros_like(v) for _ in range(world)]
        dist.all_gather(gathered, v)
        if rank == 0:
            legs.append(leg_record(cfg_, b_, steps_, world, float(tmax.item()), sum(float(g[2]) for g in gathered),
                                   [float(g[2] / g[1]) for g in gathered]))
    # teardown order, as in run_rank: rank 0's own legs (roofline, extras) run while the others wait in the final barrier;
    # nobody destroys the process group before every rank has passed it
    extras_done = 0.0
    if rank == 0:
        time.sleep(0.3)  # stands for rank 0's legs
        extras_done = time.time()
    dist.barrier()
    passed = torch.tensor([time.time(), extras_done], dtype=torch.float64)
    allp = [torch.zeros_like(passed) for _ in range(world)]
    dist.all_gather(allp, passed)
    if rank == 0:
        print(json.dumps({"dry_run": True, "backend": "gloo", "n_gpus": dist.get_world_size(), "steps": args.steps,
                          "pairs": stats["pairs"], "keypoints0": stats["keypoints0"], "matches": stats["matches"],
                          "config": {"workload": WORKLOADS[args.config][2], "pairs_per_gpu_per_step": B, "global_batch": B * world},
                          "shard_ranges": [[int(r[0]), int(r[1])] for r in ranges], "world_env": int(os.environ["WORLD_SIZE"]),
                          "scale_legs": legs, "placement": placements,
                          "teardown": {"rank0_legs_done": extras_done, "final_barrier_passed": [float(p[0]) for p in allp],
                                       "order": "final barrier after rank 0's legs, then destroy_process_group on every rank"}}))
    dist.destroy_process_group()


def _import_shard_only(name="shard"):
    """shard.py / placement.py without importing the package (which loads the HIP library and is GPU-only)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("einx_" + name, os.path.join(ROOT, "ei-nexus_official_amd", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def place_this_rank(rank, local, world):
    """Multi-rank runs: pin the rank to cores of its GPU's NUMA node and size its thread pools to that share -- BEFORE torch is
    imported and before the first GPU call (ei-nexus_official_amd/placement.py).  One-rank runs keep every core: their CPU
    baseline legs are timed on all of them."""
    if world <= 1 or os.environ.get("EINX_BENCH_NO_PLACEMENT") == "1":
        return None
    rec = _import_shard_only("placement").place_rank(local, world)
    rec["rank"] = rank
    return rec


def finish_placement(placement, torch):
    if placement is not None:
        torch.set_num_threads(placement["threads"])
        placement["torch_threads"] = torch.get_num_threads()
    return placement


def gather_placements(placement, dist, world):
    """every rank's placement record on rank 0 (a collective: all ranks call it)"""
    if placement is None or world <= 1:
        return None
    recs = [None] * world
    dist.all_gather_object(recs, placement)
    return sorted(recs, key=lambda r: r["rank"])


# ------------------------------------------------------------------------------------ workload
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


class Workload:
    """One model + one resident batch of synthetic pairs + the step function."""

    def __init__(self, pkg, dev, config, B, rank=0, calibrate=True, dense=False, log_assignment=False, ce=5, seed=11, same_scene=None):
        import torch
        self.torch, self.pkg, self.dev, self.config, self.B, self.ce = torch, pkg, dev, config, B, ce
        synth = pkg.synth
        cfg_name = WORKLOADS[config][0]
        self.cfg = cfg = pkg.default_config(cfg_name, event_channels=ce)
        self.model = model = pkg.EIM(cfg, device=dev).eval()
        self.sd = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=seed)
        # LightGlue legs run in the "same scene" regime of tests/golden/lgcal.npz (round 4): the event extractor is the image
        # extractor's twin, the events are the image + a sparse perturbation and the assignment head is calibrated -> hundreds
        # of confident matches per pair instead of ~10 with scores of 1e-6 (kernel time does not depend on it; the matcher's
        # outputs then mean something).  The MNN headline keeps the independent-networks regime of rounds 1-3.
        self.same_scene = bool(calibrate and config.endswith("_lg")) if same_scene is None else bool(same_scene)
        if self.same_scene:
            self.sd.update(synth.twin_overrides(self.sd))
        model.load_state_dict({k: torch.from_numpy(v) for k, v in self.sd.items()}, strict=False)
        for ext in (model.event_extractor.extractor, model.image_extractor.extractor):
            ext.dense_outputs = "lazy" if dense == "lazy" else bool(dense)
        model.matcher.matcher.want_log_assignment = bool(log_assignment)
        # synthetic pairs: each rank gets its own shard of the global pair index space
        self.ev_np, self.mask_np = synth.synth_events(10_000 + rank * B, B, ce)
        self.img_np = synth.synth_image(10_000 + rank * B, B)
        if self.same_scene:
            self.ev_np = synth.twin_events(self.ev_np, self.img_np)
        self.ev = torch.from_numpy(self.ev_np).to(dev)
        self.mask = torch.from_numpy(self.mask_np).to(dev)
        self.img_src = torch.from_numpy(self.img_np).to(dev)
        self.img = torch.empty_like(self.img_src)
        self.calibrated = False
        if calibrate:
            self.calibrate()

    def calibrate(self):
        """Random-weight ReLU stacks emit descriptors dominated by a per-channel constant (every keypoint
        looks alike: ~2 mutual matches in 1024).  As the golden fixtures do (tests/golden/gen_golden.py
        `calibrate`), move the per-channel spatial mean of each raw descriptor map into the last bias of its
        descriptor head, so the matcher sees descriptors that differ between keypoints.  Outside any timed
        region; the adjusted biases become part of `self.sd`, which the CPU baseline / verification use too."""
        torch, model = self.torch, self.model
        self.img.copy_(self.img_src)
        ef, imf, _ = model(self.ev, self.img, self.mask)
        msd = model.state_dict()
        over = {}
        for prefix, feats in (("event_extractor.extractor.", ef), ("image_extractor.extractor.", imf)):
            mean = feats["raw_descriptors"].mean(dim=(0, 2, 3))
            cands = [k for k in msd if k.startswith(prefix) and (k.endswith("convDb.bias") or k.endswith("_desH2.1.bias"))]
            assert len(cands) == 1, cands  # the last additive term of the descriptor head (conv bias / BatchNorm beta)
            key = cands[0]
            over[key] = (msd[key] - mean).detach().cpu()
        model.load_state_dict(over, strict=False)  # parent-level load: the extractors' native weight images are rebuilt
        for k, v in over.items():
            self.sd[k] = v.numpy()
        if self.same_scene and self.config.endswith("_lg"):
            # LightGlue's assignment head, calibrated from the final descriptors of pair 0 (synth.lightglue_calibration: the
            # rule behind tests/golden/lgcal.npz, which the reference itself ran)
            self.img.copy_(self.img_src)
            ef, imf, _ = model(self.ev, self.img, self.mask)
            one = lambda f: {"sparse_positions": f["sparse_positions"][0][None], "sparse_descriptors": f["sparse_descriptors"][0][None],  # noqa: E731
                             "image_size": [f["image_size"][0]]}
            r = model.matcher.matcher(one(ef), one(imf))
            import numpy as np
            x = np.concatenate([r["ref_descriptors0"][0, 0].cpu().numpy(), r["ref_descriptors1"][0, 0].cpu().numpy()], 0)
            lg_over, _ = self.pkg.synth.lightglue_calibration(self.sd, x, prefix="matcher.matcher.")
            model.load_state_dict({k: torch.from_numpy(v) for k, v in lg_over.items()}, strict=False)
            self.sd.update(lg_over)
        self.calibrated = True

    def step(self):
        self.img.copy_(self.img_src)  # SuperPoint scales its input in place (reference quirk), so refresh it
        return self.model(self.ev, self.img, self.mask)

    def sub(self, prefix):
        return {k[len(prefix):]: v for k, v in self.sd.items() if k.startswith(prefix)}

    def timed(self, steps, init=2, warmup=1):
        torch = self.torch
        for _ in range(init + warmup):
            self.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nm = 0
        for _ in range(steps):
            _, _, m = self.step()
            nm += sum(int(t.shape[0]) for t in m["matched_kpts0"])
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        return el / steps, nm / (steps * self.B)


def timed_stream(wl, steps, depth=2):
    """the same K forwards through EIM.forward_stream (up to `depth` batches in flight)"""
    torch = wl.torch

    def gen(n):
        for _ in range(n):
            wl.img.copy_(wl.img_src)
            yield (wl.ev, wl.img, wl.mask)

    for _ in wl.model.forward_stream(gen(3), depth=depth):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nm = 0
    for _, _, m in wl.model.forward_stream(gen(steps), depth=depth):
        nm += sum(int(t.shape[0]) for t in m["matched_kpts0"])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return el / steps, nm / (steps * wl.B)


def hip_time(torch, fn, reps, warm=2):
    """mean seconds per call, HIP events on the stream the kernels are launched on (torch's current stream)"""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def library_profile(pkg, fn):
    """{kernel class: (calls, total_ms)} of one call of fn, from the library's own HIP-event scopes."""
    import ctypes
    L = pkg.native.lib()
    L.einx_profile_enable(1)
    try:
        fn()
        buf = ctypes.create_string_buffer(1 << 16)
        L.einx_profile_report(buf, len(buf))
    finally:
        L.einx_profile_enable(0)
    out = {}
    for line in buf.value.decode().splitlines():
        name, calls, ms = line.rsplit(" ", 2)
        out[name] = (int(calls), float(ms))
    return out


def layer_table(wl):
    torch, pkg, model, B = wl.torch, wl.pkg, wl.model, wl.B
    rows = []
    for _ in range(3):  # bring the device to its working clocks before the first timed row
        wl.step()
    torch.cuda.synchronize()
    for side, ext, x0 in (("event", model.event_extractor.extractor, wl.ev), ("image", model.image_extractor.extractor, wl.img_src)):
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
                dur = hip_time(torch, lambda: layer(t_in, fold=fold), 5, warm=1)
                H_, W_ = (Hp, Wp) if fold else t_in.shape[-2:]
                fl = conv_layer_flops(layer.cin, layer.cout, layer.ks, H_, W_) * B
                rows.append((f"{side}.{cname}{li}", layer.cin, layer.cout, layer.ks, int(H_), int(W_), bool(layer.pool), round(dur * 1e6, 1),
                             round(fl / dur / 1e12, 1)))
                t_in = out
            if cname == "bb":
                feats = t_in
    for r in rows:
        print("%-14s cin=%3d cout=%3d ks=%d %3dx%3d pool=%d  %8.1f us  %6.1f TFLOP/s" % r)
    total = sum(r[7] for r in rows)
    print("total conv us (every layer as its own launch)", round(total, 1))
    # round 6: the image side's first two layers run as one launch in the pipeline (conv1ab_kernel) when the launch is large enough
    import ctypes
    ext = model.image_extractor.extractor
    eng = ext.engine()
    l0, l1 = eng.backbone[0], eng.backbone[1]
    pads = pkg.native.padder_pads(260, 346, ext.cell_size)
    Hp, Wp = 260 + pads[2] + pads[3], 346 + pads[0] + pads[1]
    L = pkg.native.lib()
    if L.einx_conv_first_two_fused_ok(ctypes.byref(l0.desc), ctypes.byref(l1.desc), B, Hp, Wp) == 1:
        Ho, Wo = (Hp // 2, Wp // 2) if l1.pool else (Hp, Wp)
        out = torch.empty((B, l1.cout, Ho, Wo), dtype=torch.float32, device=wl.dev)
        P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        st = pkg.native._stream(out)
        run = lambda: L.einx_conv_first_two_fused(P(wl.img_src), B, 260, 346, pads[2], pads[0], Hp, Wp, ctypes.byref(l0.desc), ctypes.byref(l1.desc), P(out), st)  # noqa: E731
        dur = hip_time(torch, run, 5, warm=1)
        fl = (conv_layer_flops(l0.cin, l0.cout, 3, Hp, Wp) + conv_layer_flops(l1.cin, l1.cout, 3, Hp, Wp)) * B
        two = sum(r[7] for r in rows if r[0] in ("image.bb0", "image.bb1"))
        print("%-14s cin=%3d cout=%3d ks=%d %3dx%3d pool=%d  %8.1f us  %6.1f TFLOP/s   (one launch instead of image.bb0 + image.bb1 = %.1f us)"
              % ("image.bb0+1", l0.cin, l1.cout, 3, Hp, Wp, bool(l1.pool), round(dur * 1e6, 1), round(fl / dur / 1e12, 1), two))
        print("total conv us (as launched in the pipeline)", round(total - two + dur * 1e6, 1))


def dominant_kernel_roofline(wl, value_per_gpu, kernel_only=False):
    """conv1b (64->64 3x3 at full resolution, fused ReLU + 2x2 max-pool) of the image extractor."""
    torch, pkg, model, B = wl.torch, wl.pkg, wl.model, wl.B
    ext = model.image_extractor.extractor
    eng = ext.engine()
    l0, l1 = eng.backbone[0], eng.backbone[1]
    pads = pkg.native.padder_pads(260, 346, ext.cell_size)
    Hp, Wp = 260 + pads[2] + pads[3], 346 + pads[0] + pads[1]
    fold = (pads[2], pads[0], Hp, Wp)
    x1 = l0(wl.img_src, fold=fold)
    if kernel_only:
        # no pipeline steps ran before: bring the device to its working clocks with the *first* layer's kernel, so that
        # every launch of the measured kernel in a `rocprofv3 --stats` summary of this mode is a steady-state launch
        for _ in range(300):
            l0(wl.img_src, fold=fold)
    reps = 24
    b2b = hip_time(torch, lambda: l1(x1), reps, warm=2)  # back-to-back: launch i+1 fills the CUs while launch i drains
    # per-launch duration: every launch bracketed by its OWN pair of HIP events on the launch stream, which is what a
    # rocprofv3 --kernel-trace summary of this kernel reports (begin -> end of one dispatch, ramp-up and drain included)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda.synchronize()
    for e0, e1 in evs:
        e0.record()
        l1(x1)
        e1.record()
    torch.cuda.synchronize()
    dur = sum(e0.elapsed_time(e1) for e0, e1 in evs) * 1e-3 / reps
    flops = conv_layer_flops(l1.cin, l1.cout, l1.ks, Hp, Wp) * B
    ach = flops / dur / 1e12
    traffic, traffic_src = None, None
    import glob
    pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_conv1b.json")))
    if wl.config == "sp_mnn" and B == 32 and pmcs:
        pmc = pmcs[-1]  # the latest round's counter passes
        traffic = json.load(open(pmc))["hbm_bytes_per_launch"]
        traffic_src = (f"replayed from profiles/{os.path.basename(pmc)} (FETCH_SIZE+WRITE_SIZE of separate rocprofv3 --pmc passes over this "
                       "kernel, tools/collect_profiles.sh; PMC counters cannot be read from inside this process, so this run did not re-measure them)")
    l1(x1)
    kname = pkg.native.lib().einx_conv_last_kernel().decode() + f" ({l1.cin}->{l1.cout} 3x3 @{Hp}x{Wp}, B={B})"  # what the dispatcher launched
    fused = fused_first_two_roofline(wl, l0, l1, pads, Hp, Wp, kernel_only)
    note = ("the step's two longest kernels are the two extractors' second layers: this one (event side, its own launch) and the image side's "
            "`fused_first_two_layers` launch (the same tile / K loop with the first layer recomputed inside: ~4 % longer, both layers' FLOPs "
            "counted); `roofline` stays on the plain kernel so that the figure is comparable across rounds") if fused else None
    return {"fused_first_two_layers": fused, "note": note, "kernel": kname, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "launch_ms": round(dur * 1e3, 4), "launches_timed": reps, "flop_per_launch": flops,
            "timing": "mean of per-launch HIP-event pairs (comparable with a rocprofv3 --kernel-trace average of this kernel)",
            "back_to_back_ms": round(b2b * 1e3, 4), "back_to_back_TFLOPs": round(flops / b2b / 1e12, 2),
            "hbm_frac_at_measured_rate": round(SP_PAIR_BYTES * value_per_gpu / PEAK_HBM_BYTES, 4) if wl.config == "sp_mnn" else None}


def fused_first_two_roofline(wl, l0, l1, pads, Hp, Wp, kernel_only=False):
    """Round 6: the image extractor's first two layers run as ONE launch (conv1ab_kernel: conv1a recomputed per tile on the matrix
    cores inside conv1b's launch).  Timed like the dominant kernel (per-launch HIP-event pairs on the launch stream); FLOPs = both
    layers.  None when the dispatcher would not fuse this launch (small batches)."""
    import ctypes
    torch, pkg, B = wl.torch, wl.pkg, wl.B
    L = pkg.native.lib()
    if L.einx_conv_first_two_fused_ok(ctypes.byref(l0.desc), ctypes.byref(l1.desc), B, Hp, Wp) != 1:
        return None
    Ho, Wo = (Hp // 2, Wp // 2) if l1.pool else (Hp, Wp)
    out = torch.empty((B, l1.cout, Ho, Wo), dtype=torch.float32, device=wl.dev)
    src = wl.img_src
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    st = pkg.native._stream(src)

    def run():
        rc = L.einx_conv_first_two_fused(P(src), B, 260, 346, pads[2], pads[0], Hp, Wp, ctypes.byref(l0.desc), ctypes.byref(l1.desc), P(out), st)
        assert rc == 0, L.einx_last_error()

    reps = 24
    b2b = hip_time(torch, run, reps, warm=2)
    kname = L.einx_conv_last_kernel().decode()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda.synchronize()
    for e0, e1 in evs:
        e0.record()
        run()
        e1.record()
    torch.cuda.synchronize()
    dur = sum(e0.elapsed_time(e1) for e0, e1 in evs) * 1e-3 / reps
    flops = (conv_layer_flops(l0.cin, l0.cout, l0.ks, Hp, Wp) + conv_layer_flops(l1.cin, l1.cout, l1.ks, Hp, Wp)) * B
    # the two launches it replaces, alone on the device
    x1 = l0(src, fold=(pads[2], pads[0], Hp, Wp))
    t0 = hip_time(torch, lambda: l0(src, fold=(pads[2], pads[0], Hp, Wp)), 12, warm=2)
    t1 = hip_time(torch, lambda: l1(x1), 12, warm=2)
    traffic = None
    import glob
    pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_conv1ab.json")))
    if wl.config == "sp_mnn" and B == 32 and pmcs:
        traffic = json.load(open(pmcs[-1])).get("hbm_bytes_per_launch")
    alg = B * (260 * 346 + l1.cout * Ho * Wo) * 4  # the raw image in, the second layer's output out
    return {"kernel": kname + f" (1->64->64 3x3 @{Hp}x{Wp}, B={B})", "bound": "mfma", "launch_ms": round(dur * 1e3, 4),
            "launches_timed": reps, "flop_per_launch": flops, "achieved": round(flops / dur / 1e12, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(flops / dur / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4), "back_to_back_ms": round(b2b * 1e3, 4),
            "replaces_ms": {"first_layer": round(t0 * 1e3, 4), "second_layer": round(t1 * 1e3, 4), "sum": round((t0 + t1) * 1e3, 4)},
            "algorithmic_bytes_per_launch": alg, "traffic": traffic,
            "intermediate_bytes_no_longer_moved": 2 * B * l0.cout * Hp * Wp * 4,
            "note": "image extractor, layers 1-2 as one launch; the event extractor (5 input channels) keeps two launches: its second layer is the "
                    "`roofline` kernel above"}


def stage_rooflines(wl, lg_wl=None):
    """Per-stage fractions asked for by the north_star: conv encoders (both roofs), the descriptor-correlation
    GEMM, and LightGlue's GEMM / attention kernels.  Kernel times come from the library's HIP-event scopes
    (einx_profile_*) around every launch of ONE forward with both extractors on one stream."""
    torch, pkg, model, B = wl.torch, wl.pkg, wl.model, wl.B
    out = []

    def single_stream_forward(w):
        keep = w.model.overlap_extractors
        w.model.overlap_extractors = False
        try:
            w.step()
            torch.cuda.synchronize()
        finally:
            w.model.overlap_extractors = keep

    for _ in range(5):  # the CPU baseline legs leave the device idle for ~20 s: bring it back to its working clocks first
        wl.step()
    single_stream_forward(wl)
    prof = library_profile(pkg, lambda: single_stream_forward(wl))
    if not wl.config.startswith("silk"):
        conv_ms = sum(ms for k, (c, ms) in prof.items() if k.startswith("conv_block_kernel") or k.startswith("conv1ab_kernel"))
        conv_calls = sum(c for k, (c, ms) in prof.items() if k.startswith("conv_block_kernel") or k.startswith("conv1ab_kernel"))
        fl = sp_pair_flops(wl.ce) * B
        if conv_ms > 0:
            out.append({"stage": "conv encoders + heads (both extractors, %d launches)" % conv_calls, "ms": round(conv_ms, 3),
                        "bound": "mfma", "achieved": round(fl / conv_ms / 1e9, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(fl / conv_ms / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4),
                        "hbm": {"algorithmic_bytes": SP_PAIR_BYTES * B, "achieved_GBps": round(SP_PAIR_BYTES * B / conv_ms / 1e6, 1),
                                "frac": round(SP_PAIR_BYTES * B / (conv_ms * 1e-3) / PEAK_HBM_BYTES, 4)},
                        "note": "169 FLOP/B: compute-bound, the HBM fraction is reported because the north_star names it"})
    if "mnn_tile_kernel<0>" in prof:
        c, ms = prof["mnn_tile_kernel<0>"]
        D = 128 if wl.config.startswith("silk") else 256
        fl = 2.0 * 1024 * 1024 * D * B * c
        out.append({"stage": "descriptor-correlation GEMM + fused arg-max (mnn_tile_kernel<0>)", "ms": round(ms / c, 4), "bound": "mfma",
                    "achieved": round(fl / ms / 1e9, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(fl / ms / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4)})
    tail = {k: round(ms, 4) for k, (c, ms) in prof.items() if not k.startswith("conv_block_kernel") and not k.startswith("conv1ab_kernel") and not k.startswith("mnn_tile")}
    if tail:
        out.append({"stage": "latency-bound tail (ms per forward, both sides)", "kernels_ms": tail})
    if lg_wl is not None:
        for _ in range(2):
            lg_wl.step()
        single_stream_forward(lg_wl)
        p2 = library_profile(pkg, lambda: single_stream_forward(lg_wl))
        Bl = lg_wl.B
        n = 1024
        # per pair and layer: self 2 x (Wqkv 3d^2 + ffn0 (2d)(2d)... ) see SURVEY 8d; here: measured kernel time vs the FLOPs each class executes
        d = 256
        gemm_flop_layer = 2 * (2.0 * n * d * 3 * d + 2.0 * n * 2 * d * 2 * d + 2.0 * n * 2 * d * d)  # self: qkv, ffn0 (out_proj folded), ffn3; both sides
        gemm_flop_layer += 2 * (2.0 * n * d * d * 2 + 2.0 * n * 2 * d * 2 * d + 2.0 * n * 2 * d * d)  # cross: to_qk, to_v, ffn0 (to_out folded), ffn3
        attn_flop_layer = 4 * (2.0 * n * n * d * 2)  # 2 self + 2 cross attentions: QK^T and PV
        if "lg_gemm_kernel" in p2:
            c, ms = p2["lg_gemm_kernel"]
            fl = (gemm_flop_layer * 9 + 2 * 2.0 * n * d * d) * Bl  # + final_proj on both sides
            out.append({"stage": f"LightGlue linears (lg_gemm_kernel, {c} launches, B={Bl})", "ms": round(ms, 3), "bound": "mfma",
                        "achieved": round(fl / ms / 1e9, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(fl / ms / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4)})
        for key in ("lg_attn_kernel", "lg_cross_attn_kernel"):
            if key in p2:
                c, ms = p2[key]
                share = 1.0 if ("lg_cross_attn_kernel" not in p2) else 0.5
                fl = attn_flop_layer * share * 9 * Bl
                out.append({"stage": f"LightGlue attention ({key}, {c} launches, B={Bl})", "ms": round(ms, 3), "bound": "mfma",
                            "achieved": round(fl / ms / 1e9, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(fl / ms / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4)})
        rest = {k: round(ms, 3) for k, (c, ms) in p2.items() if k.startswith("lg_") and k not in ("lg_gemm_kernel", "lg_attn_kernel", "lg_cross_attn_kernel")}
        lg_ms = sum(ms for k, (c, ms) in p2.items() if k.startswith("lg_"))
        out.append({"stage": f"LightGlue forward (all lg_* kernels, B={Bl})", "ms": round(lg_ms, 3), "bound": "mfma",
                    "achieved": round(LG_PAIR_FLOP * Bl / lg_ms / 1e9, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(LG_PAIR_FLOP * Bl / lg_ms / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4), "other_kernels_ms": rest})
    return out


def dense_stage_roofline(w):
    """The dense-output stage: upsample + L2-normalise of both sides' raw descriptor maps, a pure HBM stream."""
    torch, pkg = w.torch, w.pkg
    keep = w.model.overlap_extractors
    w.model.overlap_extractors = False  # one stream: a kernel's HIP-event pair then brackets that kernel alone
    try:
        for _ in range(3):
            w.step()
        torch.cuda.synchronize()
        prof = library_profile(pkg, lambda: (w.step(), torch.cuda.synchronize()))
    finally:
        w.model.overlap_extractors = keep
    keys = [k for k in prof if k.startswith("upsample")]
    ms = sum(prof[k][1] for k in keys)
    calls = sum(prof[k][0] for k in keys)
    by = 2.0 * w.B * 256 * 260 * 346 * 4  # both sides' [B,256,260,346] fp32 outputs; the 33x44 sources are L2-resident
    return {"stage": "dense descriptor maps (upsample + normalise, both sides, %d launches)" % calls, "ms": round(ms, 3), "bound": "hbm",
            "achieved": round(by / ms / 1e9, 3) if ms > 0 else None, "peak": PEAK_HBM_BYTES / 1e12, "unit": "TB/s",
            "frac": round(by / (ms * 1e-3) / PEAK_HBM_BYTES, 4) if ms > 0 else None, "algorithmic_bytes": by,
            "kernels_ms": {k: round(prof[k][1], 4) for k in keys}}


def host_info():
    """CPU model, logical CPUs this process may run on, physical cores behind them"""
    logical = os.cpu_count() or 1
    try:
        allowed = sorted(os.sched_getaffinity(0))
        logical = len(allowed)
    except Exception:
        allowed = None
    model, cores = None, set()
    try:
        phys = core = proc = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                proc = int(v)
            elif k == "model name" and model is None:
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and proc is not None:
                if allowed is None or proc in allowed:
                    cores.add((phys, core))
                phys = core = proc = None
    except OSError:
        pass
    physical = len(cores) if cores and (None, None) not in cores else None
    plc = _import_shard_only("placement")
    return {"cpu_model": model, "logical_cpus": logical, "physical_cores": physical, "cgroup_cpu_quota": plc.cgroup_cpu_quota(),
            "effective_cpus": plc.effective_cpus()}


def cpu_torch_protocol(wl, torch):
    """the plain-PyTorch CPU expression (oracle/torch_cpu.py) at B=1 and B=8, 2 warm-ups + 5 timed
    repeats each, median; torch threads = physical cores; B=1 once more with the reference's dense descriptor maps, which
    dominate the reference's own CPU time."""
    import statistics
    from oracle import torch_cpu
    hi = host_info()
    keep = torch.get_num_threads()
    a_ = (wl.sub("event_extractor.extractor."), wl.sub("image_extractor.extractor."))
    # thread count: the best of a short sweep around the CPUs this process can keep busy (more threads than the cgroup's quota
    # get the whole process throttled; fewer leave cores idle) -- one pair, 1 warm-up + 3 repeats per candidate
    eff = hi["effective_cpus"]
    cands = sorted({max(1, eff // 2), eff, min(max(hi["physical_cores"] or eff, eff), 2 * eff)})
    sweep = []
    for nt in cands:
        torch.set_num_threads(nt)
        run1 = lambda: torch_cpu.sp_mnn_pairs(*a_, wl.ev_np[:1], wl.mask_np[:1], wl.img_np[:1].copy(), dense=False)  # noqa: E731
        run1()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            run1()
            ts.append(time.perf_counter() - t0)
        sweep.append({"threads": nt, "median_s_per_pair": round(statistics.median(ts), 3)})
    nthreads = min(sweep, key=lambda e: e["median_s_per_pair"])["threads"]
    torch.set_num_threads(nthreads)
    legs = []
    try:
        for nb, dense, reps in ((1, False, 5), (8, False, 5), (1, True, 5)):
            nb = min(nb, wl.B)
            run = lambda: torch_cpu.sp_mnn_pairs(*a_, wl.ev_np[:nb], wl.mask_np[:nb], wl.img_np[:nb].copy(), dense=dense)  # noqa: E731
            for _ in range(2):
                run()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                run()
                ts.append(time.perf_counter() - t0)
            med = statistics.median(ts)
            legs.append({"batch": nb, "dense_outputs": dense, "median_pairs_per_s": round(nb / med, 3), "median_s": round(med, 3),
                         "min_s": round(min(ts), 3), "max_s": round(max(ts), 3), "warmups": 2, "repeats": reps})
    finally:
        torch.set_num_threads(keep)
    b8 = [l_ for l_ in legs if l_["batch"] == min(8, wl.B) and not l_["dense_outputs"]][0]
    return {"value": b8["median_pairs_per_s"], "unit": "pairs/s", "threads": nthreads, "kind": "plain PyTorch CPU expression (oracle/torch_cpu.py)",
            "sample": f"B={b8['batch']} sparse outputs, median of {b8['repeats']} after 2 warm-ups (BASELINE.md section 4)", "legs": legs,
            "thread_sweep": sweep, "thread_choice": "fastest of the sweep (candidates: half / all / twice the CPUs the cgroup's quota and the affinity mask allow)", **hi}


def harness_leg(pkg, wl, torch, steps=10, events_per_sample=60000):
    """The reference's evaluation step (test_events-image_same-time.py:130-194) end to end: raw events as the dataset hands
    them (host numpy) -> pack + H2D -> einx_voxel_grid + einx_events_mask -> EIM.forward -> MR / MMA / VDD on the device."""
    import numpy as np
    B = wl.B
    ev = pkg.SameTimeEvaluator(wl.model, wl.ce, (346, 260))
    events = [pkg.synth.synth_raw_events(5000 + b, events_per_sample) for b in range(B)]
    rep = importlib.import_module(pkg.__name__ + ".datasets.representations")

    def step():
        wl.img.copy_(wl.img_src)
        return ev.step(events, wl.img)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / steps

    # the same loop through SameTimeEvaluator.run: batch i + 1 is packed into page-locked memory, uploaded on a side stream and
    # enqueued while batch i is still on the device (every batch gets its own image tensor: SuperPoint scales it in place)
    def feed(n):
        for _ in range(n):
            yield events, wl.img_src.clone()

    for _ in ev.run(feed(3)):
        pass
    torch.cuda.synchronize()
    run_steps = max(steps, 30)  # a loop: its first batch has nothing to hide its packing and upload under (2 ms over 10 batches = 2 %)
    t0 = time.perf_counter()
    for _ in ev.run(feed(run_steps)):
        pass
    torch.cuda.synchronize()
    sec_run = (time.perf_counter() - t0) / run_steps
    # the representation alone, inputs already on the device (what the two kernels cost inside that step)
    x, y, t, p, offs = rep._pack(events, wl.dev)
    import ctypes
    L, N = pkg.native.lib(), pkg.native
    grid = torch.empty((B, wl.ce, 260, 346), dtype=torch.float32, device=wl.dev)
    mask = torch.empty((B, 1, 260, 346), dtype=torch.uint8, device=wl.dev)
    ws = torch.empty(L.einx_voxel_ws_bytes(B, wl.ce, 260, 346, int(offs[-1])), dtype=torch.uint8, device=wl.dev)
    ws2 = torch.empty(L.einx_events_ws_bytes(B, 260, 346), dtype=torch.uint8, device=wl.dev)
    op = offs.ctypes.data_as(ctypes.c_void_p)

    def dev_rep():
        L.einx_voxel_grid(N._ptr(x), N._ptr(y), N._ptr(t), N._ptr(p), op, B, wl.ce, 260, 346, 1, N._ptr(grid), N._ptr(ws), ws.numel(), N._stream(grid))
        L.einx_events_mask(N._ptr(x), N._ptr(y), op, B, 260, 346, N._ptr(ws2), N._ptr(mask), N._stream(mask))

    rep_s = hip_time(torch, dev_rep, 20)
    res = ev.result()
    streamed = {"config": wl.config, "workload": f"B{B} raw events ({events_per_sample} per sample, host numpy) -> voxel grid + events mask -> "
                + WORKLOADS[wl.config][2] + " -> MR/MMA/VDD on the device, as a LOOP with 2 batches in flight", "pairs_per_step": B,
                "value": round(B / sec_run, 2), "unit": "pairs/s", "ms_per_step": round(sec_run * 1e3, 3), "steps": run_steps,
                "h2d_bytes_per_step": int(sum(v.nbytes for e in events for v in e.values())),
                "note": "SameTimeEvaluator.run: the evaluation loop of test_events-image_same-time.py:130-194 with the next batch's host-side "
                        "packing (einx_events_pack into page-locked memory), its PCIe transfer and its two representation kernels (stage stream) and the "
                        "launches of its forward issued before the host waits for the previous batch's counts; same kernels and results as the "
                        "step-by-step leg"}
    return streamed, {"config": wl.config, "workload": f"B{B} raw events ({events_per_sample} per sample, host numpy) -> voxel grid + events mask -> "
            + WORKLOADS[wl.config][2] + " -> MR/MMA/VDD on the device", "pairs_per_step": B, "value": round(B / sec, 2), "unit": "pairs/s",
            "ms_per_step": round(sec * 1e3, 3), "steps": steps,
            "representation_ms_device": round(rep_s * 1e3, 4),
            "h2d_bytes_per_step": int(sum(v.nbytes for e in events for v in e.values())),
            "harness_metrics_mean": {k: (round(v, 5) if v == v else None) for k, v in res.items()},
            "note": "harness call pattern of the reference (test_events-image_same-time.py:130-194): includes packing the event arrays "
                    "on the host and their PCIe transfer every step (pageable memory), so it is NOT comparable with `value`; the voxel grid is "
                    "deterministic (bit-equal run to run)"}


def cpu_baseline_and_verify(wl, args, gpu_out):
    """The oracle (C port, OpenMP) on the host cores over the first pairs of the resident batch.  Its outputs
    double as the checker of the GPU outputs of the same pairs (`verified_pairs`): same weights, same inputs."""
    import numpy as np
    from oracle import oracle as orc
    cfg, B = wl.cfg, wl.B
    nb = args.cpu_pairs if args.cpu_pairs else {"sp_mnn": 32, "sp_lg": 4, "silk_mnn": 4, "silk_lg": 2}.get(wl.config, 4)
    nb = max(1, min(nb, B))
    et, it = cfg.event_extractor.type, cfg.image_extractor.type
    escale, iscale = cfg.event_extractor[et].descriptor_scale_factor, cfg.image_extractor[it].descriptor_scale_factor
    passes = 4 if (wl.config == "sp_mnn" and not args.cpu_pairs) else 1  # the headline sample: ~10-20 s of host work (128 pairs at 16 threads)
    tc = time.perf_counter()
    for _ in range(passes):
        oe = orc.extractor_forward(et, wl.sub("event_extractor.extractor."), wl.ev_np[:nb].copy(), wl.mask_np[:nb], top_k=1024, scale=escale)
        oi = orc.extractor_forward(it, wl.sub("image_extractor.extractor."), wl.img_np[:nb].copy(), None, top_k=1024, scale=iscale)
        res = []
        for b in range(nb):
            if cfg.matcher.type == "MNN":
                r = orc.mnn(oe["sparse_descriptors"][b], oi["sparse_descriptors"][b], want_la=args.log_assignment)
            else:
                r = orc.lightglue(wl.sub("matcher.matcher."), oe["sparse_positions"][b], oe["sparse_descriptors"][b],
                                  oi["sparse_positions"][b], oi["sparse_descriptors"][b])
            res.append(r)
    cpu_s = time.perf_counter() - tc
    hi = host_info()
    omp = os.environ.get("OMP_NUM_THREADS", "")
    cores = int(omp) if omp.isdigit() else hi["logical_cpus"]  # main() sizes the OpenMP pool to the CPUs the process can keep busy
    # ---- verification of the GPU outputs against the checker (outside every timed region)
    ef, imf, m = gpu_out
    verified, first_bad = 0, None
    for b in range(nb):
        ok = True
        for got, exp in ((ef, oe), (imf, oi)):
            ok &= np.array_equal(got["sparse_positions"][b].cpu().numpy(), exp["sparse_positions"][b])
            ok &= np.array_equal(got["sparse_descriptors"][b].cpu().numpy(), exp["sparse_descriptors"][b])
        g0 = m["matches0"][b].cpu().numpy().reshape(-1)
        ok &= np.array_equal(g0, np.asarray(res[b]["matches0"]).reshape(-1))
        verified += int(bool(ok))
        if not ok and first_bad is None:
            first_bad = b
    base = {"value": round(nb * passes / cpu_s, 3), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{nb * passes} pairs of the same workload through oracle/ (C, OpenMP, {cores} threads), {cpu_s:.1f} s",
            "cores_are": "OpenMP threads used = CPUs the process can keep busy (affinity mask capped by the cgroup's CPU quota: more threads "
                         "than that get the whole process throttled)", "logical_cpus": hi["logical_cpus"], "physical_cores": hi["physical_cores"],
            "cgroup_cpu_quota": hi["cgroup_cpu_quota"], "cpu_model": hi["cpu_model"], "verified_pairs": verified, "verified_of": nb,
            "verified_what": "keypoint positions+scores and descriptors bit-equal, match indices equal, GPU vs oracle on the same pairs"}
    if first_bad is not None:
        base["first_mismatch_pair"] = first_bad
    return base


def run_rank(args):
    rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    placement = place_this_rank(rank, local, world)  # affinity + thread-pool sizes, before torch / the first GPU call
    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    finish_placement(placement, torch)
    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py rank {rank}: needs a HIP device -- the product path has no CPU fallback "
                         "(use --dry-run-gloo to rehearse the launcher and the collective on CPU)")
    if local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} HIP device(s) visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    distributed = "RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ
    if os.environ.get("EINX_BENCH_NO_PG") == "1":  # tools: a launched rank without a process group (A/B of the RCCL overhead)
        distributed = False
    rccl = None
    if distributed:  # under a launcher (also with one rank): RCCL process group, one process per GPU
        dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)
        world = dist.get_world_size()
        probe = torch.ones(8, dtype=torch.float64, device=dev)
        dist.all_reduce(probe)  # communicator set-up happens here, not in the timed region
        torch.cuda.synchronize()
        assert int(probe[0].item()) == world, "RCCL all-reduce did not see every rank"
        t0 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(probe)
        torch.cuda.synchronize()
        rccl = {"backend": dist.get_backend(), "world": world, "allreduce_us": round((time.perf_counter() - t0) / 20 * 1e6, 1),
                "allreduce_payload_bytes": 64}
    placements = gather_placements(placement, dist, world) if distributed else None
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    pkg = importlib.import_module("ei-nexus_official_amd")
    B = args.batch or WORKLOADS[args.config][1]
    # --kernel-only (rocprofv3 passes over the dominant kernel): no forward at all, so that every launch of that kernel in the
    # profiler's summary is one of the steady-state launches timed here (a calibration forward would add two launches
    # that share the device with the other extractor's stream)
    wl = Workload(pkg, dev, args.config, B, rank=rank, calibrate=not (args.raw_weights or args.kernel_only), dense=args.dense,
                  log_assignment=args.log_assignment)
    model = wl.model
    acc = pkg.shard.MetricAccumulator(dev)  # pairs, keypoints(ev), keypoints(im), matches, ...
    metric_sums = torch.zeros(9, dtype=torch.float64, device=dev)
    metric_rows = []
    batch_metrics = importlib.import_module(pkg.__name__ + ".core.metrics._native_metrics").batch_metrics

    def step(accumulate=False):
        ef, imf, m = wl.step()
        if args.with_metrics:  # harness metrics of the reference's test script, computed on the device
            res = batch_metrics(ef._batched, imf._batched, model._last_match)
            if accumulate:
                metric_rows.append(res)  # [B,9] per step; summed after the timed region
        if accumulate:
            acc.add_batch(ef, imf, m)
        return ef, imf, m

    if args.layer_table:
        layer_table(wl)
        return
    if args.kernel_only:
        args.steps, args.warmup, args.no_cpu_baseline, args.no_extras = 0, 0, True, True
    else:
        # initialisation, not a measured or warm-up step: the first forward builds the kernel-native weight images
        # (repack, BN fold, LightGlue projection folding), loads the code objects, sizes the allocator pool and settles
        # the sticky NMS pass budget; do it here so that `--warmup 0` does not time a cold start.  Then forwards for
        # INIT_SECONDS: the first process on a freshly handed-over box once timed 11.4 instead of 8.4 ms per step (not
        # reproduced: tools/experiments/r6_cold_box.py reads the steady rate from the first steps on; profiles/r06_notes.md 9);
        # three seconds of untimed work are cheap insurance.  Still initialisation -- the W warm-up and the K timed steps follow
        for _ in range(2):
            step()
        t_init = time.perf_counter()
        while time.perf_counter() - t_init < INIT_SECONDS:
            step()
    for _ in range(args.warmup):
        step()

    def barrier():
        if distributed:
            dist.barrier(device_ids=[local])

    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step(accumulate=True)
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
    value = stats["pairs"] / elapsed if elapsed > 0 else 0.0

    streams = stream_overlap_report(pkg, dev) if rank == 0 else None

    # ---- all-rank legs (every N): BASELINE configs[4], SP + LightGlue at 64 pairs per GPU, sharded like the headline ----
    scale_legs, leg_wls = [], {}
    for cfg_, b_, steps_ in scale_leg_plan(args):
        rec, w_ = all_rank_leg(pkg, dev, cfg_, b_, steps_, rank, world, dist if distributed else None, local)
        if rec is not None:
            if rccl is not None:
                rec["rccl_world"] = rccl["world"]
            scale_legs.append(rec)
        if rank == 0 and world == 1:
            leg_wls[(cfg_, b_)] = w_  # reused by the per-stage rooflines below
        else:
            del w_
            torch.cuda.empty_cache()

    roofline = stages = cpu_baseline = cpu_torch = None
    extras = []
    do_extras = rank == 0 and not args.no_extras and (world == 1 or args.extras)
    if rank == 0:
        roofline = dominant_kernel_roofline(wl, value / max(world, 1), kernel_only=args.kernel_only)

    # ---- CPU baseline (the oracle: a port, not the reference files) + verification, N=1 only ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if last is None:
            last = step()
        cpu_baseline = cpu_baseline_and_verify(wl, args, last)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_cpu_torch and args.config == "sp_mnn":
        cpu_torch = cpu_torch_protocol(wl, torch)

    # ---- short extra legs: the other BASELINE configs, B=1 latency, the round-1 (un-calibrated) weights ----
    if do_extras:
        def leg(config, batch, calibrate=True, steps=3, note=None, init=2):
            w = wl if (config == args.config and batch == B and calibrate == wl.calibrated) else Workload(pkg, dev, config, batch, calibrate=calibrate)
            sec, mm = w.timed(steps, init=init)
            e = {"config": config, "workload": f"B{batch} " + WORKLOADS[config][2], "pairs_per_step": batch, "calibrated_descriptors": bool(w.calibrated),
                 "same_scene_pairs": bool(w.same_scene),
                 "value": round(batch / sec, 2), "unit": "pairs/s", "ms_per_step": round(sec * 1e3, 3), "steps": steps, "mean_matches": round(mm, 1)}
            if note:
                e["note"] = note
            extras.append(e)
            return w

        lg_wl = None
        if args.config == "sp_mnn":
            lg_wl = leg_wls.pop(("sp_lg", 64), None)
            if lg_wl is not None:  # the all-rank leg above IS the configs[3] measurement at N=1 (same timing rules)
                r_ = scale_legs[0]
                extras.append({"config": "sp_lg", "workload": r_["workload"], "pairs_per_step": 64, "calibrated_descriptors": r_.get("calibrated_descriptors"),
                               "value": r_["value"], "unit": "pairs/s", "ms_per_step": r_["ms_per_step"], "steps": r_["steps"],
                               "mean_matches": r_.get("mean_matches"), "note": "BASELINE configs[3] (= scale_legs[0])"})
            else:
                lg_wl = leg("sp_lg", 64, steps=SCALE_LEG_STEPS, note="BASELINE configs[3]")
            stages = stage_rooflines(wl, lg_wl)
            del lg_wl
            torch.cuda.empty_cache()
            w = leg("silk_mnn", 32, steps=10, note="BASELINE configs[2]", init=5)  # GB-sized activations: the allocator needs a few steps after empty_cache
            del w
            torch.cuda.empty_cache()
            w = leg("silk_lg", 32, steps=5, init=4, note="configs/model/test/EI_SiLK_LG.yaml (SiLK family + LightGlue, 128-d descriptors through input_proj)")
            del w
            torch.cuda.empty_cache()
            # the dict an unmodified reference caller gets: dense descriptor maps + dense positions + log_assignment
            w = Workload(pkg, dev, "sp_mnn", 32, dense=True, log_assignment=True)
            sec, mm = w.timed(10, init=6)  # 2 x 2.95 GB blocks per step change streams: the caching allocator settles after a few steps
            extras.append({"config": "sp_mnn", "workload": "B32 " + WORKLOADS["sp_mnn"][2], "pairs_per_step": 32, "calibrated_descriptors": True,
                           "value": round(32 / sec, 2), "unit": "pairs/s", "ms_per_step": round(sec * 1e3, 3), "steps": 10, "mean_matches": round(mm, 1),
                           "note": "reference-complete dict: dense_outputs=True (normalized_descriptors [B,256,260,346] = 2.95 GB per side, "
                                   "dense_descriptors / dense_positions) and log_assignment -- the defaults of core.modules (the headline "
                                   "materialises the sparse set only, SURVEY 8d)"})
            stages.append(dense_stage_roofline(w))
            del w
            torch.cuda.empty_cache()
            # the package default since round 4: the same dict, its dense entries computed when a caller first reads them
            w = Workload(pkg, dev, "sp_mnn", 32, dense="lazy", log_assignment=True)
            sec, mm = w.timed(10, init=4)
            extras.append({"config": "sp_mnn", "workload": "B32 " + WORKLOADS["sp_mnn"][2], "pairs_per_step": 32, "calibrated_descriptors": True,
                           "value": round(32 / sec, 2), "unit": "pairs/s", "ms_per_step": round(sec * 1e3, 3), "steps": 10, "mean_matches": round(mm, 1),
                           "note": "package default (dense_outputs='lazy' + log_assignment): every key of the reference's dict is present; "
                                   "normalized_descriptors / dense_descriptors / dense_positions are computed by the same kernels when first read "
                                   "(this leg, like the reference's evaluation scripts, never reads them; the leg above reads nothing either but "
                                   "computes them in the forward)"})
            del w
            torch.cuda.empty_cache()
            extras.extend(reversed(harness_leg(pkg, wl, torch)))  # step by step (the reference's pattern), then the streamed loop
            # single pairs (round 4 started these legs after 15 un-timed forwards because of one-off 30-80 ms host stalls; round 5
            # found the cause -- CFS throttling of the container by over-sized CPU thread pools, fixed in main() -- and removed that)
            w = leg("sp_mnn", 1, steps=50, init=2, note="single-pair latency (the reference's own call pattern, test_events-image_same-time.py:130-194): ms_per_step is ms per pair")
            # the same single pair through EIM.forward_graph: the device side of the forward captured once into a hipGraph
            w.model.forward_graph(w.ev, w.img_src, w.mask)
            for _ in range(3):
                w.model.forward_graph(w.ev, w.img_src, w.mask)
            torch.cuda.synchronize()
            tg = time.perf_counter()
            for _ in range(100):
                _, _, mg = w.model.forward_graph(w.ev, w.img_src, w.mask)
            torch.cuda.synchronize()
            sec = (time.perf_counter() - tg) / 100
            extras.append({"config": "sp_mnn", "workload": "B1 " + WORKLOADS["sp_mnn"][2], "pairs_per_step": 1, "calibrated_descriptors": True,
                           "same_scene_pairs": False, "value": round(1 / sec, 2), "unit": "pairs/s", "ms_per_step": round(sec * 1e3, 3), "steps": 100,
                           "mean_matches": float(mg["matched_kpts0"][0].shape[0]),
                           "note": "single-pair latency through EIM.forward_graph (opt-in latency mode: the ~60 launches of a forward replayed as ONE "
                                   "hipGraph launch, outputs live in graph-owned buffers until the next call; same kernels and outputs as forward)"})
            del w
            w = leg("sp_lg", 1, steps=30, init=2, note="single-pair latency with the LightGlue matcher (configs/model/test/EI_SP_LG.yaml evaluated pair by pair): ms_per_step is ms per pair")
            del w
            sec, mm = timed_stream(wl, 20)
            extras.append({"config": "sp_mnn", "workload": f"B{B} " + WORKLOADS["sp_mnn"][2], "pairs_per_step": B,
                           "calibrated_descriptors": bool(wl.calibrated), "value": round(B / sec, 2), "unit": "pairs/s",
                           "ms_per_step": round(sec * 1e3, 3), "steps": 20, "mean_matches": round(mm, 1),
                           "note": "EIM.forward_stream, 2 batches in flight: the next batch's convolutions are enqueued before the host "
                                   "waits for the previous batch's counts (same kernels and outputs; the headline above is the "
                                   "synchronous EIM.forward, one batch at a time like the reference)"})
            w = leg("sp_mnn", 32, calibrate=wl.calibrated is False, steps=10,
                    note="the other descriptor regime: " + ("calibrated" if not wl.calibrated else "round-1 un-calibrated weights (near-constant descriptors)"))
            del w
            torch.cuda.empty_cache()
        else:
            stages = stage_rooflines(wl, wl if args.config in ("sp_lg", "silk_lg") else None)

    if rank == 0:
        if rccl is not None:
            assert rccl["world"] == world == int(os.environ.get("WORLD_SIZE", world)), "the RCCL group does not span the launched ranks"
        wl_desc = f"B{B} " + WORKLOADS[args.config][2]
        out = {
            "metric": "event-image pairs/s (extract+match, 346x260, 1024 kpts)",
            "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl_desc, "pairs_per_gpu_per_step": B, "global_batch": B * world, "event_bins": wl.ce,
                       "parallelism": f"dp{world} (pairs sharded, metric all-reduce only)", "log_assignment": bool(args.log_assignment),
                       "dense_outputs": bool(args.dense),
                       "same_scene_pairs": bool(wl.same_scene),
                       "weights": "seeded synthetic, descriptor-head bias calibrated (per-channel mean removed)" if wl.calibrated
                       else "seeded synthetic, un-calibrated (near-constant descriptors)",
                       "mean_keypoints": [round(stats["keypoints0"] / pairs_total, 1), round(stats["keypoints1"] / pairs_total, 1)],
                       "mean_matches": round(stats["matches"] / pairs_total, 1),
                       "harness_metrics_mean": ([round(v, 5) for v in (metric_sums / pairs_total).tolist()] if args.with_metrics else None)},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        if scale_legs:
            out["scale_legs"] = scale_legs
        if stages:
            out["roofline_stages"] = stages
        if extras:
            out["extra_configs"] = extras
        if rccl is not None:
            out["rccl"] = rccl
        if streams is not None:
            out["streams"] = streams
        if placements is not None:
            out["placement"] = placements  # per rank: GPU, its NUMA node, the cores the rank is pinned to, thread-pool size
        if cpu_torch is not None:
            out["cpu_baseline_torch"] = cpu_torch
        print(json.dumps(out))
        sys.stdout.flush()
    if distributed:
        # rank 0's own legs (dominant-kernel roofline, extras) ran while the other ranks were already here: nobody tears the
        # communicator down before every rank has arrived
        dist.barrier(device_ids=[local])
        dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    # before numpy / torch are imported: size the CPU math libraries' thread pools to the CPUs this process can really use
    # (affinity mask capped by the cgroup's CPU quota).  Pools sized from the visible CPUs froze the whole container for
    # 30-80 ms at a time on the GPU boxes (CFS throttling; profiles/r05_notes.md), wherever the main thread happened to be.
    _import_shard_only("placement").cap_thread_pools()
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not under_launcher and (args.gpus > 1 or args.spawn):
        sys.exit(launch_ranks(args, argv))
    if args.dry_run_gloo:
        if not under_launcher:  # one-rank rehearsal
            os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        return dry_run_gloo(args)
    run_rank(args)


if __name__ == "__main__":
    main()
