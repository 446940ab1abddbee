"""Scratch: where the host time of one EIM.forward_graph call (SP+MNN, B=1) goes -- input copies, graph launch, wait, _finish."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("ei-nexus_official_amd")
dev = torch.device("cuda", 0)
w = bench.Workload(pkg, dev, "sp_mnn", 1)
m = w.model
for _ in range(50):
    m.forward_graph(w.ev, w.img_src, w.mask)
g = list(m._graphs.values())[0]
cur = torch.cuda.current_stream(dev)
N = 500
acc = [0.0] * 5
for _ in range(N):
    t0 = time.perf_counter()
    for dst, src in zip(g["inputs"], (w.ev, w.img_src, w.mask, None)):
        if dst is not None:
            dst.copy_(src, non_blocking=True)
    t1 = time.perf_counter()
    g["graph"].replay()
    t2 = time.perf_counter()
    cur.synchronize()
    t3 = time.perf_counter()
    p = g["p"]
    for bf, tmpl in ((p["ev"], g["prep_ev"]), (p["im"], g["prep_im"])):
        bf.reuse_prepared(tmpl)
    t4 = time.perf_counter()
    out = m._finish(dict(p, det_event=None, nm_event=None))
    t5 = time.perf_counter()
    for i, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5))):
        acc[i] += (b - a) * 1e6
print("us per call: copies %.1f  replay() %.1f  synchronize %.1f  reuse_prepared %.1f  _finish %.1f  sum %.1f" % (*[a / N for a in acc], sum(acc) / N))
# the graph alone, one at a time (launch -> done), vs back to back
ts = []
for _ in range(200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g["graph"].replay()
    cur.synchronize()
    ts.append((time.perf_counter() - t0) * 1e6)
ts.sort()
print("replay+sync alone: median %.1f us, min %.1f us" % (ts[len(ts) // 2], ts[0]))
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    m.forward_graph(w.ev, w.img_src, w.mask)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
