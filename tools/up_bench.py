import importlib, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("ei-nexus_official_amd")
N = pkg.native
def timed(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
raw = torch.randn(32, 256, 33, 44, device="cuda")
ms = timed(lambda: N.upsample_normalize(raw, (264, 352), (3, 3, 2, 2), 1.0))
gb = 32*256*260*346*4/1e9
print(f"upsample_normalize B=32: {ms*1e3:.0f} us  {gb/ms:.2f} TB/s written")
out = torch.empty(32, 256, 260, 346, device="cuda")
ms = timed(lambda: out.fill_(1.0)); print(f"fill_ {gb:.2f} GB: {ms*1e3:.0f} us {gb/ms:.2f} TB/s")
src = torch.empty_like(out)
ms = timed(lambda: out.copy_(src)); print(f"copy_ : {ms*1e3:.0f} us {2*gb/ms:.2f} TB/s r+w")
if len(sys.argv) > 1:
    pass
