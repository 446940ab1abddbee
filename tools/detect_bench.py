"""Per-kernel time of the detection tail (einx_score_map + einx_detect on a softmax score map of random logits) and of
einx_div_inplace at the bench shapes.   python tools/detect_bench.py [B ...]   EINX_LIB=ab_libs/libeinx_X.so: A/B build."""
import ctypes
import importlib
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("ei-nexus_official_amd")
N = pkg.native
L = N.lib()

for B in [int(a) for a in sys.argv[1:] if a.isdigit()] or [32, 1]:
    torch.manual_seed(0)
    logits = torch.randn(B, 65, 33, 44, device="cuda") * 3
    mask = torch.rand(B, 1, 260, 346, device="cuda") > 0.7
    img = torch.rand(B, 1, 260, 346, device="cuda") * 255

    def f():
        _, score = N.score_map(logits, mask, (3, 3, 2, 2), True, 4)
        N.detect(score, top_k=1024, radius=4, det_thr=1.0, pads=(3, 3, 2, 2))
        N.div_inplace(img, 1.0)

    for _ in range(3):
        f()
    torch.cuda.synchronize()
    L.einx_profile_enable(1)
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    L.einx_profile_report(buf, len(buf))
    L.einx_profile_enable(0)
    print(f"B={B}:", end="")
    for line in buf.value.decode().splitlines():
        name, calls, t = line.rsplit(" ", 2)
        print(f"  {name} {float(t) / int(calls) * 1e3:.1f} us x{int(calls) // 10}", end="")
    print()
