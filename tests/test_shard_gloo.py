"""world_size-2 gloo test of the multi-GPU scheme (shard pairs, all-reduce metric accumulators)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT, load_pkg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    pkg = load_pkg()
    dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
    r, lr, w = pkg.shard.env_ranks()
    assert (r, lr, w) == (rank, rank, world)
    lo, hi = pkg.shard.shard_range(total, rank, world)
    acc = pkg.shard.MetricAccumulator("cpu")
    for i in range(lo, hi):  # fake per-pair results derived from the global pair index
        acc.add(1, 1000 + i, 900 + 2 * i, i % 7, 0.5 * i)
    acc.all_reduce()
    if rank == 0:
        torch.save(acc.as_dict(), out)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_allreduce_world2(tmp_path):
    total, world = 37, 2
    out = str(tmp_path / "acc.pt")
    mp.spawn(_worker, args=(world, _free_port(), total, out), nprocs=world, join=True)
    got = torch.load(out)
    idx = range(total)
    assert got == {"pairs": float(total), "keypoints0": float(sum(1000 + i for i in idx)), "keypoints1": float(sum(900 + 2 * i for i in idx)),
                   "matches": float(sum(i % 7 for i in idx)), "match_score_sum": float(sum(0.5 * i for i in idx))}


def test_shard_ranges_cover_everything():
    pkg = load_pkg()
    for total in (0, 1, 7, 32, 512, 1001):
        for world in (1, 2, 3, 8):
            spans = [pkg.shard.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
