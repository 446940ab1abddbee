#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py ei-nexus_official_amd/csrc/conv.hip [filter-substring] [-D...]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flt = [a for a in sys.argv[2:] if not a.startswith("-")]
    extra = [a for a in sys.argv[2:] if a.startswith("-")]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            name = t.split(":", 1)[1].strip()
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            cur = {"name": dem.replace("(anonymous namespace)::", "")}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    print("%-90s %5s %5s %5s %6s %4s %6s" % ("kernel", "VGPR", "AGPR", "SGPR", "LDS", "occ", "spill"))
    for r in rows:
        if flt and not all(f in r["name"] for f in flt):
            continue
        print("%-90s %5s %5s %5s %6s %4s %6s" % (r["name"][:90], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"),
                                                 r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]"), r.get("VGPRs Spill")))


if __name__ == "__main__":
    main()
