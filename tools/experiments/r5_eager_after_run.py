"""Eager single-pair time before / after the harness legs (bench: 0.76 -> 1.06 ms after SameTimeEvaluator.run)."""
import gc
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

bench._import_shard_only("placement").cap_thread_pools()
import torch  # noqa: E402

pkg = importlib.import_module("ei-nexus_official_amd")
dev = torch.device("cuda", 0)
wl1 = bench.Workload(pkg, dev, "sp_mnn", 1)
wl = bench.Workload(pkg, dev, "sp_mnn", 32)


def t(tag):
    sec, _ = wl1.timed(200, init=5)
    print(f"{tag:60s} {sec * 1e3:.3f} ms   reserved {torch.cuda.memory_reserved() >> 20} MiB", flush=True)


t("baseline")
ev = pkg.SameTimeEvaluator(wl.model, wl.ce, (346, 260))
events = [pkg.synth.synth_raw_events(5000 + b, 60000) for b in range(32)]
for _ in range(5):
    wl.img.copy_(wl.img_src)
    ev.step(events, wl.img)
torch.cuda.synchronize()
t("after 5 x step()")
mode = sys.argv[1] if len(sys.argv) > 1 else "run2"
if mode == "run1":
    for _ in ev.run(((events, wl.img_src.clone()) for _ in range(5)), depth=1):
        pass
elif mode == "fs":
    def gen(n):
        for _ in range(n):
            yield (wl.ev, wl.img_src.clone(), wl.mask)
    for _ in wl.model.forward_stream(gen(5), depth=2):
        pass
else:
    for _ in ev.run(((events, wl.img_src.clone()) for _ in range(5)), depth=2):
        pass
torch.cuda.synchronize()
t(f"after 5 batches through {mode}")
ev._stages = {}
gc.collect()
torch.cuda.empty_cache()
t("after dropping the stages + empty_cache")
