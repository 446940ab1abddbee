#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per counter set, as tools/experiments/r3_pmc_conv.sh writes them) into one
per-kernel-variant table: LDS bank-conflict ratio, LDS issue stalls, MFMA-busy and the effective clock.

    python tools/pmc_tiles.py gpurun_out/pmc_conv3 [name-filter] > profiles/r03_pmc_conv_tiles.json

effective MHz = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs) / kernel duration (guide: DVFS give-back).
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    root = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else "conv_block_kernel"
    acc = collections.defaultdict(lambda: collections.defaultdict(float))  # kernel -> counter -> sum over launches
    cdur = collections.defaultdict(lambda: collections.defaultdict(float))  # kernel -> counter -> summed ns of the pass that carried it
    dur = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
        p = os.path.basename(os.path.dirname(f))
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if flt not in k:
                continue
            k = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cdur[k][r["Counter_Name"]] += d
            key = (r["Dispatch_Id"], k)
            if key not in seen:
                seen.add(key)
                dur[k][p] += d
                launches[k][p] += 1
    out = {}
    for k, v in acc.items():
        row = {"launches_per_pass": max(launches[k].values())}
        mean_us = [dur[k][p] / launches[k][p] / 1e3 for p in dur[k]]
        row["mean_us_per_launch_profiled"] = round(sum(mean_us) / len(mean_us), 1)
        if v.get("SQ_LDS_IDX_ACTIVE"):
            row["lds_bank_conflict_per_idx_active"] = round(v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], 4)
        if v.get("SQ_WAVE_CYCLES"):
            row["wait_inst_lds_per_wave_cycle"] = round(v.get("SQ_WAIT_INST_LDS", 0) / v["SQ_WAVE_CYCLES"], 4)
            row["wait_inst_any_per_wave_cycle"] = round(v.get("SQ_WAIT_INST_ANY", 0) / v["SQ_WAVE_CYCLES"], 4)
        if v.get("GRBM_GUI_ACTIVE"):
            cyc = v["GRBM_GUI_ACTIVE"] / 8
            row["mfma_busy_frac"] = round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024), 4)
            row["effective_mhz"] = round(cyc / cdur[k]["GRBM_GUI_ACTIVE"] * 1e3, 0)
        if v.get("SQ_INSTS_MFMA"):
            row["valu_per_mfma"] = round(v.get("SQ_INSTS_VALU", 0) / v["SQ_INSTS_MFMA"], 3)
            row["lds_per_mfma"] = round(v.get("SQ_INSTS_LDS", 0) / v["SQ_INSTS_MFMA"], 3)
        out[k] = row
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
