"""round 5: the one-off 30-80 ms host stalls in the first forwards after a LightGlue weight reload (profiles/r04_notes.md 5).
One fresh process per variant:  python tools/experiments/r5_stall_hunt.py VARIANT
  base        as bench.py's single-pair SP+LightGlue leg (same-scene regime: LightGlue weights reloaded by calibrate())
  noreload    independent-networks regime (no LightGlue reload)
  emptycache  torch.cuda.empty_cache() + synchronize after the reload
  gc          gc.collect() + gc.freeze() after the reload
  nowatch     EINX_NO_WATCH=1 (no weight watch launches / read-backs)
  sleep       2 s of host sleep after the reload (is it time-based?)
  threads1 / passive   CPU math libraries without worker threads / with sleeping instead of spinning workers
  plainreload the LightGlue weights loaded again 2 s later (no host-side linear algebra in front of the loop)
  blasonly    no reload, host-side linear algebra right before the loop
prints the wall time of each of the first 40 forwards, allocator counters around the slow ones."""
import gc, importlib, os, statistics, sys, time
variant = sys.argv[1] if len(sys.argv) > 1 else "base"
if variant == "nowatch":
    os.environ["EINX_NO_WATCH"] = "1"
if variant == "threads1":  # no worker threads in the CPU math libraries (numpy / torch CPU ops of calibrate())
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = "1"
if variant == "passive":   # worker threads sleep instead of spinning after a parallel region
    os.environ.update(OMP_WAIT_POLICY="PASSIVE", GOMP_SPINCOUNT="0", KMP_BLOCKTIME="0", OPENBLAS_THREAD_TIMEOUT="1")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("ei-nexus_official_amd")
dev = torch.device("cuda", 0)
w = bench.Workload(pkg, dev, "sp_lg", 1, same_scene=(variant != "noreload"))
torch.cuda.synchronize()
if variant == "plainreload":
    # the reload WITHOUT the host-side linear algebra in front of it: wait until the calibration's aftermath is over, then load
    # the same LightGlue weights again (what a user does: load_state_dict, then evaluate pair by pair)
    time.sleep(2.0)
    sdm = {k: torch.from_numpy(v) for k, v in w.sd.items() if k.startswith("matcher.")}
    w.model.load_state_dict(sdm, strict=False)
if variant == "blasonly":
    # no reload at all, only host-side linear algebra right before the loop
    time.sleep(2.0)
    import numpy as np
    a = np.random.rand(2048, 256).astype(np.float32)
    for _ in range(3):
        np.linalg.svd(a, full_matrices=False)
        (a.T @ a).sum()
    t_ = torch.rand(2048, 2048)
    (t_ @ t_).sum().item()
if variant == "emptycache":
    torch.cuda.empty_cache(); torch.cuda.synchronize()
if variant == "gc":
    gc.collect(); gc.freeze()
if variant == "sleep":
    time.sleep(2.0)
keys = ("num_device_alloc", "num_device_free", "num_alloc_retries", "reserved_bytes.all.current", "segment.all.current")
def snap():
    s = torch.cuda.memory_stats(dev)
    return tuple(s.get(k, 0) for k in keys)
def cg():
    """cgroup CPU bandwidth: (quota string, nr_throttled, throttled microseconds)"""
    out = {}
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        try:
            out[os.path.basename(path)] = open(path).read().strip()
        except OSError:
            pass
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for line in open(path):
                k, v = line.split()
                if "throttled" in k:
                    out[k] = int(v)
            break
        except OSError:
            pass
    return out
cg0 = cg()
ts, snaps = [], [snap()]
if variant == "gcdisable":
    gc.disable()
gc_log, gc_t0 = [], [0.0]
def gc_cb(phase, info):
    if phase == "start":
        gc_t0[0] = time.perf_counter()
    else:
        gc_log.append((info["generation"], (time.perf_counter() - gc_t0[0]) * 1e3, info["collected"], len(ts)))
gc.callbacks.append(gc_cb)
if os.environ.get("EINX_STALL_MARK"):
    print("MARK loop start", file=sys.stderr, flush=True)
for i in range(40):
    t0 = time.perf_counter()
    w.step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
    snaps.append(snap())
    if os.environ.get("EINX_STALL_MARK"):
        print(f"MARK forward {i} took {ts[-1]:.2f} ms", file=sys.stderr, flush=True)
med = statistics.median(ts)
slow = [(i, round(t, 1)) for i, t in enumerate(ts) if t > 3 * med]
cg1 = cg()
print("   cgroup", {k: (v if not isinstance(v, int) else v - cg0.get(k, 0)) for k, v in cg1.items()}, "(throttle counters: change over the 40 forwards); affinity",
      len(os.sched_getaffinity(0)), "cpus")
print(f"{variant}: median {med:.3f} ms, forwards slower than 3x the median: {slow}; collections > 2 ms (generation, ms, collected, during forward): "
      f"{[(g, round(ms, 1), c, i) for g, ms, c, i in gc_log if ms > 2]}; objects tracked {len(gc.get_objects())}")
for i, _ in slow:
    print("   forward", i, "allocator before/after", dict(zip(keys, snaps[i])), dict(zip(keys, snaps[i + 1])))
