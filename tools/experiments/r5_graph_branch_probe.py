"""Scratch: when does the second branch of a captured two-branch graph start?  Two streams, K small kernels each (x.add_(1) on 1 MB),
captured with torch.cuda.graph and replayed; run under `rocprofv3 --kernel-trace`, then r5_graph_timeline_report.py prints the timeline.
argv[1] = K (default 30), argv[2] = 'zero' to put a cudaMemsetAsync-like node (tensor.zero_ on a byte tensor) in the middle of branch 1."""
import sys, time
import torch
K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
mid = len(sys.argv) > 2 and sys.argv[2] == "zero"
dev = torch.device("cuda", 0)
x = torch.zeros(1 << 18, device=dev)
y = torch.zeros(1 << 18, device=dev)
z = torch.zeros(1 << 12, device=dev, dtype=torch.uint8)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def body():
    cur = torch.cuda.current_stream()
    x.mul_(1.0)
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        for i in range(K):
            y.add_(1.0)
    for i in range(K):
        x.add_(1.0)
        if mid and i == K // 2:
            z.zero_()
    cur.wait_stream(s2)
    x.add_(y)


with torch.cuda.stream(s1):
    for _ in range(3):
        body()
s1.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s1):
    body()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    g.replay()
e1.record()
torch.cuda.synchronize()
print(f"K={K} mid_memset={mid}: replay back to back {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per graph ({2 * K + 2} kernels)")
time.sleep(0.05)
g.replay()
torch.cuda.synchronize()
