"""events -> voxel grid / events mask kernels at B=32 x 60k events, device-resident inputs (tuning aid)"""
import ctypes, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("ei-nexus_official_amd")
N = pkg.native
from importlib import import_module
check = import_module(pkg.__name__ + "._lib").check
B, n, H, W, bins = 32, 60000, 260, 346, 5
g = torch.Generator(device="cuda").manual_seed(0)
x = (torch.rand(B * n, device="cuda", generator=g) * (W - 1)).floor()
y = (torch.rand(B * n, device="cuda", generator=g) * (H - 1)).floor()
t = torch.rand(B * n, device="cuda", generator=g, dtype=torch.float64).sort().values
p = (torch.rand(B * n, device="cuda", generator=g) > 0.5).float()
offs = np.arange(B + 1, dtype=np.int64) * n
L = N.lib()
grid = torch.empty((B, bins, H, W), device="cuda")
mask = torch.empty((B, 1, H, W), dtype=torch.uint8, device="cuda")
ws = torch.empty(L.einx_events_ws_bytes(B, H, W), dtype=torch.uint8, device="cuda")
wsv = torch.empty(L.einx_voxel_ws_bytes(B, bins, H, W, B * n), dtype=torch.uint8, device="cuda")
def vg():
    check(L.einx_voxel_grid(N._ptr(x), N._ptr(y), N._ptr(t), N._ptr(p), offs.ctypes.data_as(ctypes.c_void_p), B, bins, H, W, 1, N._ptr(grid), N._ptr(wsv), wsv.numel(), N._stream(grid)), "vg")
def mk():
    check(L.einx_events_mask(N._ptr(x), N._ptr(y), offs.ctypes.data_as(ctypes.c_void_p), B, H, W, N._ptr(ws), N._ptr(mask), N._stream(mask)), "mask")
def timed(f, k=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
a, b = timed(vg), timed(mk)
print(f"voxel grid  B={B} x {n} events -> [{B},{bins},{H},{W}]: {a*1e3:.0f} us  ({B*n/a/1e6:.1f} G events/s, {B/a*1e3:.0f} samples/s)")
print(f"events mask B={B} x {n} events: {b*1e3:.0f} us")
