"""Scratch: which part of the B=1 forward survives hipGraph capture?  Each case runs in its own process."""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = ["fwd_batched_nofork", "enqueue_nofork", "fwd_batched"]


def run(case):
    sys.path.insert(0, ROOT)
    import torch
    import bench
    pkg = importlib.import_module("ei-nexus_official_amd")
    dev = torch.device("cuda", 0)
    w = bench.Workload(pkg, dev, "sp_mnn", 1)
    m = w.model
    if case == "fwd_batched_nooverlap":
        m.overlap_extractors = False
    ext = m.image_extractor
    st = torch.cuda.Stream()
    img = w.img_src.clone()

    def body():
        if case.startswith("extract"):
            return ext.extract_batched(img, None)
        if case == "enqueue_nofork":
            return m._enqueue(w.ev, img, w.mask, slot="g")
        if case.startswith("fwd_batched"):
            return m.forward_batched(w.ev, img, w.mask)
        if case == "enqueue_noevents":
            ev, im, mr = m.forward_batched(w.ev, img, w.mask)
            ev.prepare(); im.prepare()
            return ev, im, mr
        return m._enqueue(w.ev, img, w.mask, slot="g")

    with torch.cuda.stream(st):
        for _ in range(3):
            img.copy_(w.img_src)
            body()
        img.copy_(w.img_src)
    st.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        out = body()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("CASE", case, "captured and replayed OK")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for c in CASES:
            env = dict(os.environ)
            if c.endswith("nofork"):
                env["EINX_NO_FORK"] = "1"
            r = subprocess.run([sys.executable, os.path.abspath(__file__), c], env=env, capture_output=True, text=True)
            tail = [l for l in (r.stdout + r.stderr).splitlines() if l.strip()][-2:]
            print(c, "rc", r.returncode, "|", " / ".join(tail)[:300])
