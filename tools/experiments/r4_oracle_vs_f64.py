"""Scratch: CPU oracle LightGlue against the reference's fp32 and float64 results on the lgcal / lg fixtures."""
import sys, os, json
ROOT = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import Golden, lg_inputs, state_dict_for, lg_noise
from oracle import oracle as orc
LG = Golden("lg"); Z = np.load(os.path.join(ROOT, "tests/golden/lgcal.npz"))
for name in ("d256", "d128"):
    c = dict(LG.cases[name]); c["state_keys"] = json.loads(bytes(LG[f"{name}.state_keys"]).decode())
    sd = state_dict_for(c)
    d0, d1, k0, k1 = lg_inputs(c)
    r = orc.lightglue(sd, k0, d0, k1, d1)
    la = r["log_assignment"].astype(np.float64)
    print(name, "oracle-ref32", np.abs(la - LG[f"{name}.la"][0]).max(), "oracle-f64", np.abs(la - Z[f"lg.{name}.la_f64"]).max(), "ref32-f64", lg_noise(f"lg.{name}")["la_f64"])
from helpers import twin_state_dict_for, twin_inputs, sub_dict, split
LGCAL = Golden("lgcal")
import test_oracle_golden as T
for name in LGCAL.cases:
    c = LGCAL.cases[name]
    ef, imf, sd = T._run_twin_case(orc, c)
    ms = T._match_lists(orc, c["cfg"], sd, ef, imf)
    for b, r in enumerate(ms):
        la = r["log_assignment"].astype(np.float64)[::31, ::29]
        e32 = np.abs(la - LGCAL[f"{name}.m.la_probe2"][b]); e64 = np.abs(la - LGCAL[f"{name}.m.la_probe2_f64.{b}"])
        r32 = np.abs(LGCAL[f"{name}.m.la_probe2"][b].astype(np.float64) - LGCAL[f"{name}.m.la_probe2_f64.{b}"])
        ms0 = r["matching_scores0"].astype(np.float64)
        print(name, b, "la: oracle-ref32", e32.max(), "oracle-f64", e64.max(), "ref32-f64 (probe)", r32.max(), "(all)", lg_noise(f"{name}.{b}")["la_f64"],
              "| ms: oracle-f64", np.abs(ms0 - LGCAL[f"{name}.m.matching_scores0_f64.{b}"]).max(), "ref32-f64", lg_noise(f"{name}.{b}")["ms_f64"])
        i = np.unravel_index(e64.argmax(), e64.shape); print("   worst at", i, "value", la[i])
