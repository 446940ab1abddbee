"""round 5: the first layers' time is bimodal from process to process (image 130 / 157 us, event 232 / 246 us).  Does it follow the
ADDRESS of the 761 MB output tensor?  Time both layers with the output placed at different offsets inside one big buffer."""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("ei-nexus_official_amd")
N = pkg.native
dev = torch.device("cuda", 0)
w = bench.Workload(pkg, dev, "sp_mnn", 32)
for _ in range(3): w.step()
L = N.lib()
n_out = 32 * 64 * 264 * 352
big = torch.empty(n_out + (64 << 20), dtype=torch.float32, device=dev)  # + 256 MB of slack
print("base address %#x" % big.data_ptr())
for side, ext, x in (("image", w.model.image_extractor.extractor, w.img_src), ("event", w.model.event_extractor.extractor, w.ev)):
    layer = ext.engine().backbone[0]
    B, _, Hs, Ws = x.shape
    for off_bytes in (0, 256, 4096, 65536, 1 << 20, (1 << 21), (1 << 21) + 4096, 3 << 20, 1 << 24, (1 << 24) + (1 << 21), 1 << 26, (1 << 26) + 128 * 1024):
        out = big[off_bytes // 4: off_bytes // 4 + n_out]
        fn = lambda: N.check(L.einx_conv_block(N._ptr(x), B, Hs, Ws, 2, 3, 264, 352, ctypes.byref(layer.desc), N._ptr(out), N._stream(x)), "conv")
        t = bench.hip_time(torch, fn, 10, warm=3)
        print(f"{side} first layer, output at base + {off_bytes:>10d} B (addr % 2MiB = {(out.data_ptr() % (1 << 21)):>8d}): {t * 1e6:7.1f} us")
