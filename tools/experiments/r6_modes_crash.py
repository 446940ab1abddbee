"""fuzz_parity's modes cases only (forward_graph capture + replays, forward_stream, dense modes), seed by seed, with the seed
being run left in a trail file: bisecting a crash in hipGraph replay that appeared with the probed side streams."""
import faulthandler, os, sys, time
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuzz_parity as F
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120
trail = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/modes_trail.txt"
t0, seed, n = time.time(), 660002, 0
while time.time() - t0 < seconds:
    with open(trail, "w") as fh:
        fh.write(f"{seed} {n}\n")
    F.modes_case(seed, ["sp", "silk"])
    seed += 4
    n += 1
print("modes cases clean:", n, flush=True)
