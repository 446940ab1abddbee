"""Multi-GPU scheme of the hot path: pairs are independent (BatchNorm in eval mode, per-sample
matcher calls -- reference core/modules/Matchers.py:192-201), so a batch of pairs is sharded over
ranks with NO data-path collective.  The only exchange is one all-reduce (SUM) of a small fp64
vector of metric accumulators -- RCCL over xGMI on MI355X (`backend="nccl"`), gloo in CPU tests.
One process per GPU, launched by torchrun (RANK / LOCAL_RANK / WORLD_SIZE, env:// rendezvous as in
the reference's train_extractor.py:82-91)."""
import os

import torch
import torch.distributed as dist

FIELDS = ("pairs", "keypoints0", "keypoints1", "matches", "match_score_sum")


def env_ranks():
    """(rank, local_rank, world_size) from the torchrun environment."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(total, rank, world):
    """Contiguous block of `total` pair indices owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(total), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class MetricAccumulator:
    """fp64 sums of per-pair statistics; `all_reduce()` is the job's one collective."""

    def __init__(self, device="cpu"):
        self.device = device
        self.host = [0.0] * len(FIELDS)  # accumulated on the host; one device tensor at reduce time
        self.v = None

    def add_batch(self, events_feats, image_feats, matches):
        n0 = sum(int(p.shape[0]) for p in events_feats["sparse_positions"])
        n1 = sum(int(p.shape[0]) for p in image_feats["sparse_positions"])
        nm = sum(int(t.shape[0]) for t in matches["matched_kpts0"]) if matches is not None else 0
        self.add(len(events_feats["sparse_positions"]), n0, n1, nm, 0.0)

    def add(self, pairs, k0, k1, matches, score_sum):
        for i, x in enumerate((pairs, k0, k1, matches, score_sum)):
            self.host[i] += float(x)

    def all_reduce(self):
        self.v = torch.tensor(self.host, dtype=torch.float64, device=self.device)
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(self.v, op=dist.ReduceOp.SUM)
        return self

    def as_dict(self):
        vals = self.v.tolist() if self.v is not None else self.host
        return {k: float(x) for k, x in zip(FIELDS, vals)}
