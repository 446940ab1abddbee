"""bench.py as its own rank launcher (`python bench.py --gpus N` spawns N ranks before any GPU call):
rehearsed on CPU with gloo through the same launcher code, and the no-GPU failure mode is loud."""
import json
import os
import subprocess
import sys

import pytest
import torch

from helpers import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(*args, env_drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=240, env=env)


def test_launcher_spawns_two_gloo_ranks():
    r = _run("--gpus", "2", "--dry-run-gloo", "--steps", "4")
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # only rank 0 prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True
    B, steps = 32, 4
    assert d["pairs"] == 2 * steps * B  # whole-job aggregate over both ranks (the accumulator all-reduce)
    assert d["keypoints0"] == steps * (B * 1000 + 0) + steps * (B * 1000 + 1)
    assert d["matches"] == steps * 10 + steps * 20
    assert d["config"]["global_batch"] == 2 * B
    # the scaling run must carry BASELINE configs[4] next to the headline at every N: SP+LightGlue, 64 pairs per GPU, all ranks
    assert "MNN" in d["config"]["workload"]
    legs = d["scale_legs"]
    assert len(legs) == 1 and legs[0]["config"] == "sp_lg" and "LightGlue" in legs[0]["workload"]
    assert legs[0]["pairs_per_gpu_per_step"] == 64 and legs[0]["global_batch"] == 128 and legs[0]["n_gpus"] == 2
    assert legs[0]["steps"] >= 10 and legs[0]["value"] > 0
    pr = legs[0]["per_rank_pairs_per_s"]
    assert 0 < pr["min"] <= pr["max"]  # one entry per rank went into the spread (rank 1 sleeps longer in the rehearsal)
    assert pr["min"] < pr["max"]


def test_scale_legs_can_be_switched_off_and_are_not_doubled_for_sp_lg():
    r = _run("--gpus", "2", "--dry-run-gloo", "--steps", "1", "--no-scale-legs")
    assert r.returncode == 0, r.stderr
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])["scale_legs"] == []


def test_launcher_passes_config_and_batch():
    r = _run("--gpus", "2", "--dry-run-gloo", "--steps", "1", "--config", "sp_lg", "--batch", "64")
    assert r.returncode == 0, r.stderr
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["config"]["pairs_per_gpu_per_step"] == 64 and d["config"]["global_batch"] == 128
    assert "LightGlue" in d["config"]["workload"]
    assert d["scale_legs"] == []  # the headline already is the configs[4] workload


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a box without a GPU")
def test_without_gpu_every_rank_fails_loudly_and_nothing_hangs():
    r = _run("--gpus", "2", "--steps", "1")
    assert r.returncode != 0
    assert "rank 0" in r.stderr and "rank 1" in r.stderr and "HIP device" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]  # no bench line without a measurement


def test_rank_env_is_honoured_without_relaunch():
    """Under torchrun (RANK/WORLD_SIZE already set) bench.py must be a rank, not a launcher."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-run-gloo", "--steps", "2"], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["pairs"] == 64


def test_launcher_world_8_on_gloo_shards_512_pairs():
    """BASELINE configs[4] as the driver launches it (N = 8, 64 pairs per GPU): the launcher, the rendezvous, the accumulator
    all-reduce, the all-rank leg and the shard arithmetic at world 8 -- on gloo, no GPU (the 8-GPU run itself is the driver's)."""
    r = _run("--gpus", "8", "--dry-run-gloo", "--steps", "2", "--config", "sp_lg")
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 == d["world_env"]
    assert d["config"]["pairs_per_gpu_per_step"] == 64 and d["config"]["global_batch"] == 512
    assert d["pairs"] == 8 * 2 * 64
    assert d["matches"] == 2 * 10 * sum(range(1, 9))
    assert d["shard_ranges"] == [[64 * k, 64 * (k + 1)] for k in range(8)]  # contiguous, disjoint, covering 0..512
    r = _run("--gpus", "8", "--dry-run-gloo", "--steps", "1")
    assert r.returncode == 0, r.stderr
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    leg = d["scale_legs"][0]
    assert leg["n_gpus"] == 8 and leg["global_batch"] == 512 and leg["pairs_per_gpu_per_step"] == 64
    assert leg["per_rank_pairs_per_s"]["min"] < leg["per_rank_pairs_per_s"]["max"]


def test_world_8_placement_record_and_teardown_order():
    """Round 5: every rank is pinned before its first GPU call (cores of its GPU's NUMA node, or an even split of the allowed
    CPUs where the topology is not readable, as in this container), sizes its thread pools to that share, and the record is
    in rank 0's line; no rank destroys the process group before rank 0's own legs are done (final barrier)."""
    r = _run("--gpus", "8", "--dry-run-gloo", "--steps", "1")
    assert r.returncode == 0, r.stderr
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    pl = d["placement"]
    assert [p["rank"] for p in pl] == list(range(8)) and [p["gpu"] for p in pl] == list(range(8))
    sys.path.insert(0, os.path.join(ROOT, "ei-nexus_official_amd"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("einx_placement", os.path.join(ROOT, "ei-nexus_official_amd", "placement.py"))
    plc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(plc)
    allowed = sorted(os.sched_getaffinity(0))
    sets = [set(plc.parse_cpulist(p["cpus"])) for p in pl]
    for p, cs in zip(pl, sets):
        assert cs and cs <= set(allowed) and p["pinned"] is True
        assert p["n_cpus"] == len(cs) and 1 <= p["threads"] <= len(cs) and p["torch_threads"] == p["threads"]
    if len(allowed) >= 8:
        assert all(not (a & b) for i, a in enumerate(sets) for b in sets[i + 1:]), "ranks share cores"
    td = d["teardown"]
    assert len(td["final_barrier_passed"]) == 8
    assert min(td["final_barrier_passed"]) >= td["rank0_legs_done"] > 0  # everyone left the barrier after rank 0's legs
    # one rank: no pinning (its CPU baseline legs use every core)
    r = _run("--gpus", "1", "--dry-run-gloo", "--steps", "1")
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])["placement"] is None


def test_placement_plan_follows_the_gpu_numa_nodes(tmp_path):
    """placement.plan on a fake sysfs: 4 GPUs on 2 NUMA nodes x 8 cores x 2 threads -> each rank gets half of its GPU's node,
    whole cores (SMT siblings together), disjoint from its neighbour's."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("einx_placement", os.path.join(ROOT, "ei-nexus_official_amd", "placement.py"))
    plc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(plc)
    root = tmp_path
    ncpu = 32  # cpu c and c + 16 are SMT siblings; node 0: cores 0-7, node 1: cores 8-15
    for c in range(ncpu):
        d = root / f"devices/system/cpu/cpu{c}/topology"
        d.mkdir(parents=True)
        (d / "thread_siblings_list").write_text(f"{c % 16},{c % 16 + 16}\n")
    nodes = {0: "0-7,16-23", 1: "8-15,24-31"}
    for n, lst in nodes.items():
        d = root / f"devices/system/node/node{n}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(lst + "\n")
    (root / "class/kfd/kfd/topology/nodes/0").mkdir(parents=True)
    (root / "class/kfd/kfd/topology/nodes/0/properties").write_text("cpu_cores_count 16\nsimd_count 0\n")
    for g in range(4):
        d = root / f"class/kfd/kfd/topology/nodes/{g + 1}"
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {128 + g}\n")
        dev = root / f"class/drm/renderD{128 + g}/device"
        dev.mkdir(parents=True)
        (dev / "numa_node").write_text(f"{g // 2}\n")
        (dev / "local_cpulist").write_text(nodes[g // 2] + "\n")
    env = {k: os.environ.pop(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES") if k in os.environ}
    try:
        plans = [plc.plan(g, 4, allowed=list(range(ncpu)), sys_root=str(root)) for g in range(4)]
    finally:
        os.environ.update(env)
    assert [p["numa_node"] for p in plans] == [0, 0, 1, 1]
    assert [p["cpus"] for p in plans] == ["0-3,16-19", "4-7,20-23", "8-11,24-27", "12-15,28-31"]
    assert all(p["physical_cores"] == 4 and p["threads"] == 4 for p in plans)
    # a restricted affinity mask is respected; no topology -> even split
    p = plc.plan(1, 4, allowed=[0, 1, 2, 3, 4, 5, 16, 17], sys_root=str(root))
    assert set(plc.parse_cpulist(p["cpus"])) <= {0, 1, 2, 3, 4, 5, 16, 17}
    q = [plc.plan(g, 4, allowed=list(range(8)), sys_root=str(tmp_path / "nothing")) for g in range(4)]
    assert [x["cpus"] for x in q] == ["0-1", "2-3", "4-5", "6-7"] and all(x["numa_node"] == -1 for x in q)
    # a cgroup CPU quota is bandwidth shared by the ranks: each pool is sized to its share minus two CPUs' worth (the Python thread
    # and the HIP runtime's helpers run beside the pool; at exactly the quota the container still gets throttled -- round 5)
    (root / "fs/cgroup").mkdir(parents=True)
    (root / "fs/cgroup/cpu.max").write_text("1600000 100000\n")
    assert plc.cgroup_cpu_quota(str(root)) == 16.0
    assert [plc.plan(g, 4, allowed=list(range(ncpu)), sys_root=str(root))["threads"] for g in range(4)] == [2, 2, 2, 2]  # share 4 -> 2
    assert plc.plan(0, 1, allowed=list(range(ncpu)), sys_root=str(root))["threads"] == 8  # one rank: its node's 8 cores < 16 - 2
    (root / "fs/cgroup/cpu.max").write_text("800000 100000\n")
    assert plc.plan(0, 1, allowed=list(range(ncpu)), sys_root=str(root))["threads"] == 6  # 8 cores, quota 8 -> 6
    (root / "fs/cgroup/cpu.max").write_text("1600000 100000\n")
    assert plc.pool_threads(str(root)) == min(len(os.sched_getaffinity(0)), 16) - (2 if len(os.sched_getaffinity(0)) >= 16 else 0)
    (root / "fs/cgroup/cpu.max").write_text("max 100000\n")
    assert plc.cgroup_cpu_quota(str(root)) is None and plc.pool_threads(str(root)) == plc.effective_cpus(str(root))


def test_dead_rank_at_world_8_stops_the_others_quickly():
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["EINX_BENCH_DRYRUN_FAIL_RANK"] = "5"
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-run-gloo", "--steps", "1"], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 7, (r.returncode, r.stderr[-500:])
    assert "rank 5 exited with status 7" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert time.time() - t0 < 120  # the other seven ranks were stopped, nobody waits for the rendezvous timeout


def test_gpu_max_hw_queues_defaults_inside_ranks():
    """RCCL's streams take hardware queues; ranks default GPU_MAX_HW_QUEUES to 8 unless the caller set it."""
    code = "import os, sys; sys.argv=['bench.py']; import importlib.util as u; s=u.spec_from_file_location('b', %r); m=u.module_from_spec(s); s.loader.exec_module(m); print(os.environ['GPU_MAX_HW_QUEUES'])" % BENCH
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120).stdout.strip() == "8"
    env["GPU_MAX_HW_QUEUES"] = "2"
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120).stdout.strip() == "2"


def test_placement_divides_by_the_ranks_on_this_host(monkeypatch):
    """on a multi-node torchrun (2 x 8) the per-host divisions must use LOCAL_WORLD_SIZE, not the global WORLD_SIZE."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("einx_placement", os.path.join(ROOT, "ei-nexus_official_amd", "placement.py"))
    plc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(plc)
    monkeypatch.setenv("WORLD_SIZE", "16")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert plc.local_world() == 8 and plc.local_world(16) == 8
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    assert plc.local_world() == 16 and plc.local_world(4) == 4
    monkeypatch.delenv("WORLD_SIZE")
    assert plc.local_world() == 1
