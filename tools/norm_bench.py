"""time einx_normalize_map (+channels-last copy) and einx_desc_sample at the bench shape (tuning aid)"""
import importlib, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("ei-nexus_official_amd")
N = pkg.native
def timed(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
raw = torch.randn(32, 256, 33, 44, device="cuda")
print(f"normalize_map+cl B=32: {timed(lambda: N.normalize_map(raw, 1.0, want_cl=True)):.1f} us")
co, cl = N.normalize_map(raw, 1.0, want_cl=True)
idx = torch.randint(0, 264 * 352, (32, 1024), device="cuda", dtype=torch.int32).sort(dim=1).values.contiguous()
cnt = torch.full((32,), 1024, dtype=torch.int32, device="cuda")
print(f"desc_sample (cl)     : {timed(lambda: N.desc_sample(raw, idx, cnt, (264, 352), True, 1.0, raw_cl=cl)):.1f} us")
logits = torch.randn(32, 65, 33, 44, device="cuda")
print(f"score_map            : {timed(lambda: N.score_map(logits, None, (3, 3, 2, 2), border=4)):.1f} us")
