#!/usr/bin/env python3
"""Find global loads that hipcc serialised: compiles every csrc/*.hip to gfx950 assembly (device only, no GPU needed) and lists
  (a) innermost loops with at most 3 global loads and an `s_waitcnt vmcnt(0)` (one memory round trip per iteration), and
  (b) kernels in which at least 4 `vmcnt(0)` waits cover at most 2 loads each (guarded per-element loads in straight-line code).
A hit is a candidate, not a verdict: cold paths (partial tiles, fall-backs) show up too.   python tools/isa_audit.py"""
import glob
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "ei-nexus_official_amd", "csrc")
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math --cuda-device-only -S".split()
LOAD = re.compile(r"\b(global_load|buffer_load|flat_load)")
WAIT0 = re.compile(r"s_waitcnt.*vmcnt\(0\)")


def demangle(name):
    return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:100]


def functions(path):
    cur, out = None, {}
    for ln in open(path).read().split("\n"):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif cur:
            out[cur].append(ln)
            if "s_endpgm" in ln:
                cur = None
    return out


def main():
    tmp = tempfile.mkdtemp(prefix="einx_isa_")
    for hip in sorted(glob.glob(os.path.join(SRC, "*.hip"))):
        base = os.path.basename(hip)[:-4]
        asm = os.path.join(tmp, base + ".s")
        extra = ["-fno-slp-vectorize"] if base == "desc" else []  # as the Makefile
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + [hip, "-o", asm], check=True, stderr=subprocess.DEVNULL)
        for fn, body in functions(asm).items():
            labels = {m.group(1): i for i, ln in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):.*Loop Header", ln)] if m}
            for lab, start in labels.items():
                end = next((i for i in range(len(body) - 1, start, -1) if re.search(r"s_c?branch\w*\s+" + re.escape(lab) + r"\b", body[i])), None)
                if end is None or any(start < j < end for j in labels.values()):
                    continue  # not a loop, or not innermost
                seg = body[start:end + 1]
                nload = sum(1 for ln in seg if LOAD.search(ln))
                if 0 < nload <= 3 and any(WAIT0.search(ln) for ln in seg):
                    print(f"(a) {base:10s} {demangle(fn):100s} loop {lab}: {nload} load(s) per iteration, vmcnt(0) inside")
            since = small = waits = 0
            for ln in body:
                if LOAD.search(ln):
                    since += 1
                elif WAIT0.search(ln):
                    if since:
                        waits += 1
                        small += since <= 2
                    since = 0
            if small >= 4:
                print(f"(b) {base:10s} {demangle(fn):100s} {small} of {waits} vmcnt(0) waits cover <= 2 loads")


if __name__ == "__main__":
    main()
