#!/usr/bin/env python3
"""What the reference does with a pooled VGGExtractor(padding=0) (build container only):
    python tests/golden/gen_pad0_pooled.py   ->  tests/golden/pad0_pooled.json
The class constructs, and every forward raises IndexError: with a mask at `score[~score_mask] = 0` (EventExtractors.py:553-554), without
one in filter_sparse_feats on the dense outputs (:608 -> :499-500).  The exception types and messages are recorded per input size."""
import importlib.util, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(HERE, "gen_golden.py"))
g = importlib.util.module_from_spec(spec)
spec.loader.exec_module(g)  # stubs the optional imports and puts /root/reference on the path
import torch  # noqa: E402
from core.modules.event_extractors.EventExtractors import VGGExtractor  # noqa: E402

out = {"torch": torch.__version__, "cases": []}
m = VGGExtractor(in_channels=5, feat_channels=128, descriptor_dim=256, nms_radius=4, detection_top_k=50, detection_threshold=1.0, padding=0).eval()
out["state_keys"] = len(m.state_dict())
for B, H, W in ((1, 120, 152), (2, 123, 150), (1, 260, 346)):
    ev, mask = g.synth.synth_events(7, B, 5, H, W)
    rec = {"B": B, "H": H, "W": W}
    for tag, args in (("no_mask", (torch.from_numpy(ev),)), ("mask", (torch.from_numpy(ev), torch.from_numpy(mask)))):
        try:
            with torch.no_grad():
                m(*args)
            rec[tag] = None
        except Exception as e:  # noqa: BLE001
            rec[tag] = {"type": type(e).__name__, "message": str(e)}
    out["cases"].append(rec)
    print(rec)
json.dump(out, open(os.path.join(HERE, "pad0_pooled.json"), "w"), indent=1)
