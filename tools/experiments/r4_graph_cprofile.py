import cProfile, importlib, os, pstats, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("ei-nexus_official_amd")
dev = torch.device("cuda", 0)
w = bench.Workload(pkg, dev, "sp_mnn", 1)
fn = lambda: w.model.forward_graph(w.ev, w.img_src, w.mask)
for _ in range(30):
    fn()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    fn()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
