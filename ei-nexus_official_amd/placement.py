"""Host-side placement of one rank process per GPU (multi-GPU runs: bench.py, harness users).

Pairs are sharded over ranks with no data-path collective (shard.py), so what eight ranks on one node compete for is the
HOST: every rank enqueues ~60 launches per forward from Python and builds its output lists on the CPU.  Before a rank makes
its first GPU call it is pinned to cores of its GPU's NUMA node (the node's cores are divided among the ranks whose GPUs sit
there, SMT siblings kept together) and its thread pools are sized to that share -- the reference's launcher leaves both to
the OS (train_extractor.py:82-91 only reads RANK / LOCAL_RANK / WORLD_SIZE).

Pure Python on sysfs (no torch, no HIP call): importable before anything touches the GPU.  Where the topology cannot be
read (containers without /sys/class/kfd) the allowed CPUs are split evenly by local rank."""
import glob
import os
import re


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in (text or "").replace("\n", "").split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def format_cpulist(cpus):
    cpus = sorted(set(cpus))
    runs, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        runs.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(runs)


def cgroup_cpu_quota(sys_root="/sys"):
    """CPUs' worth of CFS bandwidth this process's cgroup may use per period (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us`), or
    None when unlimited / unreadable.  Round 5: the GPU boxes hand out 16 CPUs of quota while 256 logical CPUs are visible;
    thread pools sized from the visible CPUs (OpenMP 128, OpenBLAS 64) spin-wait after every parallel region, burn the quota
    within milliseconds and the kernel then FREEZES EVERY THREAD of the cgroup until the next 100 ms period -- the 30-80 ms
    "host stalls inside hipLaunchKernel" of round 4 (profiles/r05_notes.md)."""
    txt = _read(os.path.join(sys_root, "fs/cgroup/cpu.max"))
    if txt:
        parts = txt.split()
        if len(parts) == 2 and parts[0] != "max" and float(parts[1]) > 0:
            return float(parts[0]) / float(parts[1])
        return None
    q, per = _read(os.path.join(sys_root, "fs/cgroup/cpu/cpu.cfs_quota_us")), _read(os.path.join(sys_root, "fs/cgroup/cpu/cpu.cfs_period_us"))
    if q and per and float(q) > 0 and float(per) > 0:
        return float(q) / float(per)
    return None


def effective_cpus(sys_root="/sys"):
    """CPUs this process can actually keep busy: its affinity mask capped by the cgroup's CPU bandwidth quota"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    q = cgroup_cpu_quota(sys_root)
    if q is not None:
        n = min(n, max(1, int(q)))
    return max(1, n)


def pool_threads(sys_root="/sys"):
    """Default size of the CPU math libraries' pools: effective_cpus() minus two when the limit is a cgroup quota.  The quota is
    bandwidth, not cores: a pool of exactly quota threads that spin-wait after a parallel region plus the Python thread and the
    HIP runtime's helper threads still overdraws it, and the container is frozen for the rest of the period (measured on a
    16-CPU quota: pools of 16 -> 3 throttled periods and 6-8 ms hiccups in 160 forwards, pools of 14 -> none;
    tools/experiments/r5_stall_hunt5.py, profiles/r05_notes.md 2)."""
    n = effective_cpus(sys_root)
    q = cgroup_cpu_quota(sys_root)
    if q is not None and n >= 4 and n >= int(q):
        n -= 2
    return max(1, n)


def cap_thread_pools(n=None):
    """Size the CPU math libraries' pools (OpenMP / MKL / OpenBLAS: torch's CPU operators, numpy) to `n` threads (default:
    pool_threads()) through their environment variables -- call before they are imported.  Existing settings are kept when
    they are not larger."""
    n = pool_threads() if n is None else max(1, int(n))
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        cur = os.environ.get(var)
        if not (cur and cur.isdigit() and 0 < int(cur) <= n):
            os.environ[var] = str(n)
    return n


def gpu_topology(sys_root="/sys"):
    """[(numa_node, [cpus of that node])] for every GPU in KFD topology order (the order HIP enumerates devices in);
    [] when the topology is not readable."""
    gpus = []
    nodes = glob.glob(os.path.join(sys_root, "class/kfd/kfd/topology/nodes/*"))
    for node in sorted(nodes, key=lambda p: int(os.path.basename(p)) if os.path.basename(p).isdigit() else 1 << 30):
        props = _read(os.path.join(node, "properties"))
        if not props:
            continue
        kv = dict(line.split(None, 1) for line in props.splitlines() if len(line.split(None, 1)) == 2)
        if int(kv.get("simd_count", "0")) <= 0:
            continue  # a CPU node
        minor = kv.get("drm_render_minor")
        dev = os.path.join(sys_root, f"class/drm/renderD{minor}/device") if minor else None
        numa = _read(os.path.join(dev, "numa_node")) if dev else None
        cpus = parse_cpulist(_read(os.path.join(dev, "local_cpulist"))) if dev else []
        numa = int(numa) if numa is not None and re.fullmatch(r"-?\d+", numa) else -1
        if numa >= 0 and not cpus:
            cpus = parse_cpulist(_read(os.path.join(sys_root, f"devices/system/node/node{numa}/cpulist")))
        gpus.append((numa, cpus))
    return gpus


def _visible(n_gpus):
    """physical GPU index of every visible device (simple integer lists in HIP_/ROCR_/CUDA_VISIBLE_DEVICES only)"""
    idx = list(range(n_gpus))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        val = os.environ.get(var)
        if val and re.fullmatch(r"\s*\d+(\s*,\s*\d+)*\s*", val):
            pick = [int(x) for x in val.split(",")]
            idx = [idx[i] for i in pick if i < len(idx)]
    return idx


def _core_of(cpu, sys_root="/sys"):
    sib = _read(os.path.join(sys_root, f"devices/system/cpu/cpu{cpu}/topology/thread_siblings_list"))
    lst = parse_cpulist(sib) if sib else [cpu]
    return min(lst) if lst else cpu


def plan(local_rank, world, allowed=None, sys_root="/sys"):
    """{gpu, numa_node, cpus, physical_cores, threads, source} for the rank that drives local GPU `local_rank` of the `world`
    ranks ON THIS HOST (placement.local_world: LOCAL_WORLD_SIZE, not the global WORLD_SIZE).  `allowed`: CPUs this process may use
    (default: its current affinity mask).  The GPU -> NUMA node map assumes that HIP enumerates devices in KFD topology order
    (true on the single-vendor nodes this targets, not guaranteed by ROCm): the record says so in `source`, best effort."""
    if allowed is None:
        try:
            allowed = sorted(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            allowed = list(range(os.cpu_count() or 1))
    allowed = sorted(allowed)
    world = max(1, int(world))
    topo = gpu_topology(sys_root)
    vis = _visible(len(topo))
    node_of = {}  # local index -> numa node
    for loc in range(min(world, len(vis)) if vis else 0):  # (ranks beyond the visible GPUs have no node)
        node_of[loc] = topo[vis[loc]][0]
    my_node = node_of.get(local_rank, -1)
    source = "even split of the allowed CPUs (GPU topology not readable)"
    cpus = None
    if my_node >= 0:
        node_cpus = [c for c in topo[vis[local_rank]][1] if c in set(allowed)]
        peers = sorted(loc for loc, nd in node_of.items() if nd == my_node)
        if node_cpus:
            # whole physical cores, in core order, divided among the ranks whose GPUs sit on this node
            by_core = {}
            for c in node_cpus:
                by_core.setdefault(_core_of(c, sys_root), []).append(c)
            cores = sorted(by_core)
            k, n = peers.index(local_rank), len(peers)
            lo, hi = k * len(cores) // n, (k + 1) * len(cores) // n
            if hi > lo:
                cpus = sorted(c for core in cores[lo:hi] for c in by_core[core])
                source = f"NUMA node {my_node} of the GPU (KFD topology order assumed = HIP device order), share {k + 1} of {n}"
    if not cpus:
        lo, hi = local_rank * len(allowed) // world, (local_rank + 1) * len(allowed) // world
        cpus = allowed[lo:hi] if hi > lo else [allowed[local_rank % len(allowed)]]
    phys = len({_core_of(c, sys_root) for c in cpus})
    threads = max(1, phys)
    quota = cgroup_cpu_quota(sys_root)
    if quota is not None:  # the ranks of this host share the cgroup's CPU bandwidth; two CPUs' worth stays free per rank (pool_threads)
        share = int(quota) // world
        threads = max(1, min(threads, share - 2 if share >= 4 else share))
    return {"gpu": int(local_rank), "numa_node": int(my_node), "cpus": format_cpulist(cpus), "n_cpus": len(cpus), "physical_cores": phys,
            "threads": threads, "cgroup_cpu_quota": quota, "source": source}


def apply(p):
    """Pin this process to the plan's CPUs and size the host thread pools (call before the first GPU call and, for the
    OpenMP pools to see it, before importing torch).  Returns the plan with what actually took effect."""
    cpus = parse_cpulist(p["cpus"])
    try:
        os.sched_setaffinity(0, cpus)
        p = dict(p, pinned=True)
    except (AttributeError, OSError) as e:  # not fatal: an unpinned rank is slower, not wrong
        p = dict(p, pinned=False, pin_error=str(e))
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ[var] = str(p["threads"])
    return p


def local_world(world=None):
    """ranks on THIS host: LOCAL_WORLD_SIZE (torchrun sets it), else the given / global world size (single-node launchers).
    Every per-host division (cores of a NUMA node, the cgroup's CPU quota) is by this number, not by the global WORLD_SIZE: on a
    2 x 8 torchrun the global figure would pin every rank to 1/16 of its node's cores."""
    v = os.environ.get("LOCAL_WORLD_SIZE")
    if v and v.isdigit() and int(v) > 0:
        return int(v)
    if world is None:
        v = os.environ.get("WORLD_SIZE")
        world = int(v) if v and v.isdigit() else 1
    return max(1, int(world))


def place_rank(local_rank, world=None):
    """`world`: ranks on this host (default: LOCAL_WORLD_SIZE, else WORLD_SIZE)."""
    return apply(plan(local_rank, local_world(world)))
