"""Which case of `fuzz_parity.py --large --seed0 400000` stops making progress after seed 400014?  One line per case, flushed to
gpurun_out/ BEFORE and AFTER it runs (run under `timeout`):  timeout -k 5 140 python tools/experiments/r5_large_seed_probe.py 400015 400022"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_parity as fp  # noqa: E402

fp.LARGE = True
out = open(os.path.join(ROOT, "gpurun_out", "s2_large_probe.log"), "a")
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    out.write(f"start {seed}\n")
    out.flush()
    os.fsync(out.fileno())
    t0 = time.time()
    try:
        d = fp.one_case(seed, ("sp", "silk"))
        out.write(f"done  {seed} {time.time() - t0:.1f}s {d}\n")
    except Exception as e:  # noqa: BLE001
        out.write(f"FAIL  {seed} {time.time() - t0:.1f}s {type(e).__name__}: {str(e)[:300]}\n")
    out.flush()
    os.fsync(out.fileno())
