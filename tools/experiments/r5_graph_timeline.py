"""Scratch: a few forward_graph replays (SP+MNN, B=1) for a rocprofv3 kernel trace; tools/experiments/r5_graph_timeline_report.py
turns the trace of the LAST replay into a timeline (kernel, start offset, duration, gap to the previous kernel on the same queue)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("ei-nexus_official_amd")
dev = torch.device("cuda", 0)
w = bench.Workload(pkg, dev, "sp_mnn", 1)
for _ in range(30):
    w.model.forward_graph(w.ev, w.img_src, w.mask)
torch.cuda.synchronize()
time.sleep(0.05)
w.model.forward_graph(w.ev, w.img_src, w.mask)
torch.cuda.synchronize()
