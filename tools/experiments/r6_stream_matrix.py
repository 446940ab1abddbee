"""Which streams of a process run side by side?  Pairwise einx_stream_overlap_us over the default stream, twelve streams made
with hipStreamCreateWithFlags and a few from torch's pool, alone and after a one-rank process group has been set up
(`--pg`).  Prints the matrix of elapsed/spin ratios (1 = overlap, 2 = serialised)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import importlib, torch
pkg = importlib.import_module("ei-nexus_official_amd")
from importlib import import_module
N = import_module("ei-nexus_official_amd._native")
lib = N.lib()
pg = "--pg" in sys.argv
torch.cuda.set_device(0)
if pg:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    t = torch.ones(16, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
hip = ctypes.CDLL("libamdhip64.so")
streams = [("null", None)]
for i in range(12):
    h = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(h), 1) == 0
    streams.append((f"h{i}", h.value))
for i in range(4):
    s = torch.cuda.Stream()
    streams.append((f"t{i}", s.cuda_stream)); streams[-1] += (s,)
SPIN = 300
def probe(a, b):
    out = ctypes.c_float()
    rc = lib.einx_stream_overlap_us(ctypes.c_void_p(a), ctypes.c_void_p(b), SPIN, ctypes.byref(out))
    assert rc == 0, rc
    return out.value / SPIN
probe(None, streams[1][1])
print("mode", "process group" if pg else "plain", "GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))
names = [s[0] for s in streams]
print("      " + " ".join(f"{n:>4}" for n in names))
for i, si in enumerate(streams):
    row = []
    for j, sj in enumerate(streams):
        row.append(" .  " if j < i else f"{probe(si[1], sj[1]):4.1f}")
    print(f"{si[0]:>5} " + " ".join(row))
