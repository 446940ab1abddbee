#!/usr/bin/env python3
"""Fixture of the event representation's degenerate inputs, generated from the reference (build container only):
    python tests/golden/gen_events_degenerate.py   ->  tests/golden/events_degenerate.npz
All time stamps equal (one event; a burst with one stamp) make t_norm = 0 / 0 = NaN in
/root/reference/datasets/representations.py:76-80; torch's `.int()` maps NaN to INT_MIN on the CPU, so the range mask (:94-101)
drops every event and the grid stays zero.  Two distinct stamps among many equal ones are the neighbouring, non-degenerate case.
The inputs are stored with the outputs; nothing of the reference's source is."""
import importlib.util, os, sys, types
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
for name in ("cv2", "h5py", "hdf5plugin", "numba", "tqdm"):
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)  # imported at module level by the reference, unused by this function
spec = importlib.util.spec_from_file_location("ref_representations", "/root/reference/datasets/representations.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)
torch.set_num_threads(1)
rng = np.random.default_rng(20261004)
out = {"meta_torch": np.array(torch.__version__)}
cases = {
    "one_event": dict(n=1, stamps="equal", size=(2, 20, 30)),
    "burst_one_stamp": dict(n=500, stamps="equal", size=(5, 40, 50)),
    "two_stamps": dict(n=500, stamps="two", size=(5, 40, 50)),
    "one_event_one_bin": dict(n=1, stamps="equal", size=(1, 12, 9)),
}
for name, c in cases.items():
    n, (bins, H, W) = c["n"], c["size"]
    ev = {"x": rng.uniform(-1, W + 1, n).astype(np.float32), "y": rng.uniform(-1, H + 1, n).astype(np.float32),
          "t": np.full(n, 1.5e9), "p": rng.choice([0.0, 1.0], n).astype(np.float32)}
    if c["stamps"] == "two":
        ev["t"][n // 2:] += 0.02
    for norm in (False, True):
        grid = ref.events_to_voxel_grid({k: v.copy() for k, v in ev.items()}, (bins, H, W), normalize=norm).numpy()
        out[f"{name}.grid_norm{int(norm)}"] = grid
    for k, v in ev.items():
        out[f"{name}.{k}"] = v
    out[f"{name}.size"] = np.array([bins, H, W])
np.savez_compressed(os.path.join(HERE, "events_degenerate.npz"), **out)
print({k: (v.shape, float(np.abs(v).sum())) for k, v in out.items() if "grid" in k})
