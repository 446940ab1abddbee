"""time einx_detect alone on a synthetic peaky score map (tuning aid)"""
import importlib, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("ei-nexus_official_amd")
N = pkg.native
torch.manual_seed(0)
s = (torch.rand(32, 1, 264, 352, device="cuda") ** 8).contiguous()
s[:, :, :4] = 0; s[:, :, -4:] = 0; s[:, :, :, :4] = 0; s[:, :, :, -4:] = 0
f = lambda: N.detect(s, top_k=1024, radius=4, det_thr=1.0, pads=(3, 3, 2, 2), nms_iters=8)
d = f(); torch.cuda.synchronize()
print("not converged:", int(d.not_converged.sum()), "counts", int(d.counts.min()), int(d.counts.max()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print(f"detect (8 NMS passes + select) B=32: {e0.elapsed_time(e1)/20*1e3:.0f} us")
