"""conv1b (64->64 3x3 @264x352, pooled) with / without the BatchNorm epilogue on dense and on ReLU-sparse inputs (tuning aid)"""
import importlib, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("ei-nexus_official_amd")
N = pkg.native
def timed(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
torch.manual_seed(0)
w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
b = torch.randn(64, device="cuda") * 0.1
bn = (torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.1, torch.randn(64, device="cuda") * 0.1, torch.rand(64, device="cuda") + 0.5, 1e-5)
plain = N.ConvLayer(w, b, None, relu=True, pool=True)
withbn = N.ConvLayer(w, b, bn, relu=True, pool=True)
dense = torch.randn(32, 64, 264, 352, device="cuda")
sparse = torch.relu(dense)
zeros = torch.zeros_like(dense)
flop = 2 * 64 * 64 * 9 * 264 * 352 * 32
for name, x in (("dense", dense), ("relu-sparse", sparse), ("zeros", zeros)):
    for lname, layer in (("no BN", plain), ("BN", withbn)):
        ms = timed(lambda: layer(x))
        print(f"{name:12s} {lname:6s} {ms*1e3:8.1f} us  {flop/ms/1e9:6.1f} TFLOP/s")
