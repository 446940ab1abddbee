"""Dense descriptor maps alone (einx_upsample_normalize at B=32, 256 x 33x44 -> 260x346): total and per-kernel time.
EINX_LIB=ab_libs/libeinx_X.so selects an A/B build."""
import ctypes, importlib, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("ei-nexus_official_amd")
N = pkg.native
L = N.lib()
def timed(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
nums = [a for a in sys.argv[1:] if a.isdigit()]
B = int(nums[0]) if nums else 32
raw = torch.randn(B, 256, 33, 44, device="cuda")
f = lambda: N.upsample_normalize(raw, (264, 352), (3, 3, 2, 2), 1.0)
ms = timed(f)
gb = B*256*260*346*4/1e9
print(f"upsample_normalize B={B}: {ms*1e3:.0f} us  {gb/ms:.2f} TB/s written", end="  |")
L.einx_profile_enable(1)
for _ in range(5): f()
torch.cuda.synchronize()
buf = ctypes.create_string_buffer(1 << 16)
L.einx_profile_report(buf, len(buf))
L.einx_profile_enable(0)
for line in buf.value.decode().splitlines():
    name, calls, t = line.rsplit(" ", 2)
    print(f"  {name} {float(t)/int(calls)*1e3:.0f} us", end="")
print()
if "--ref" in sys.argv:
    out = torch.empty(B, 256, 260, 346, device="cuda")
    ms = timed(lambda: out.fill_(1.0)); print(f"fill_ {gb:.2f} GB: {ms*1e3:.0f} us {gb/ms:.2f} TB/s")
