"""The harness leg alone: step by step vs SameTimeEvaluator.run (2 batches in flight).
    gpurun -- python tools/experiments/r5_harness_stream.py"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

bench._import_shard_only("placement").cap_thread_pools()
import torch  # noqa: E402

pkg = importlib.import_module("ei-nexus_official_amd")
dev = torch.device("cuda", 0)
wl = bench.Workload(pkg, dev, "sp_mnn", 32)
for leg in bench.harness_leg(pkg, wl, torch):
    print(json.dumps({k: v for k, v in leg.items() if k not in ("note", "harness_metrics_mean")}))
