"""round 5: does it pay to run the two extractors' convolutions one after the other, with the first extractor's latency-bound
tail (score, NMS passes, selection, sampling) under the second one's convolutions, instead of interleaving both and ending
with both tails at once?  Op-level path (the handle-level call has no hook between convolutions and tail)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("ei-nexus_official_amd")
dev = torch.device("cuda", 0)
w = bench.Workload(pkg, dev, "sp_mnn", 32)
m = w.model
def timed(n=40):
    for _ in range(5): w.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): w.step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("handle-level, interleaved (shipped): %.3f ms" % timed())
ee, ie = m.event_extractor.extractor.engine(), m.image_extractor.extractor.engine()
ee.use_handle = ie.use_handle = False
print("op-level, interleaved:               %.3f ms" % timed())
ev_done = torch.cuda.Event()
side = m._side_stream(dev)
ee.after_convs = lambda: ev_done.record(torch.cuda.current_stream(dev))
orig = m.image_extractor.extract_batched
def staged(*a, **k):
    torch.cuda.current_stream(dev).wait_event(ev_done)  # the image side's convolutions start when the event side's are done
    return orig(*a, **k)
m.image_extractor.extract_batched = staged
print("op-level, event convs -> [event tail || image convs] -> image tail: %.3f ms" % timed())
