#!/usr/bin/env python3
"""gpurun_out/ev/ (tools/collect_profiles.sh) -> profiles/rNN_*: bench lines, rocprofv3 per-kernel stats and the
PMC summary of the dominant kernel.  Usage: python tools/assemble_profiles.py [round-prefix, default r01]"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EV = os.path.join(ROOT, "gpurun_out", "ev")
PR = os.path.join(ROOT, "profiles")
R = sys.argv[1] if len(sys.argv) > 1 else "r06"


def last_json_line(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"no JSON line in {path}")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"einx_match::", "", name)
    return name


def main():
    os.makedirs(PR, exist_ok=True)
    for src, dst in (("bench_sp_mnn.json", "bench_sp_mnn_b32.json"), ("bench_sp_mnn_full.json", "bench_sp_mnn_b32_full_dict.json"),
                     ("bench_sp_mnn_metrics.json", "bench_sp_mnn_b32_with_metrics.json"), ("bench_sp_lg.json", "bench_sp_lg_b64.json"),
                     ("bench_silk.json", "bench_silk_mnn_b32.json"), ("bench_silk_lg.json", "bench_silk_lg_b32.json")):
        if not os.path.exists(os.path.join(EV, src)):
            continue
        j = last_json_line(os.path.join(EV, src))
        json.dump(j, open(os.path.join(PR, f"{R}_{dst}"), "w"), indent=1)
        print(dst, j["value"], j["unit"], "roofline", j.get("roofline", {}).get("achieved"))
    for src, dst in (("bench_spawn1", "rccl_one_rank_launcher.log"), ("bench_torchrun1", "rccl_one_rank_torchrun.log")):
        if os.path.exists(os.path.join(EV, src + ".json")):
            with open(os.path.join(PR, f"{R}_{dst}"), "w") as fo:
                fo.write("# " + ("python bench.py --gpus 1 --spawn --no-cpu-baseline --no-extras" if "spawn" in src else
                                 "python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 ... bench.py --gpus 1 "
                                 "--no-cpu-baseline --no-extras") + "\n")
                fo.write(open(os.path.join(EV, src + ".err")).read())
                fo.write(open(os.path.join(EV, src + ".json")).read())
    for src, dst in (("layer_table.txt", "conv_layer_table.txt"), ("latency_b1.txt", "latency_b1.txt"), ("latency_b1_lg.txt", "latency_b1_sp_lg.txt"),
                     ("latency_graph.txt", "latency_graph.txt"), ("events_bench.txt", "events_bench.txt")):
        if os.path.exists(os.path.join(EV, src)):
            txt = "\n".join(ln for ln in open(os.path.join(EV, src)).read().splitlines() if "amdgpu.ids" not in ln)
            open(os.path.join(PR, f"{R}_{dst}"), "w").write(txt + "\n")
    if os.path.exists(os.path.join(EV, "up_bench.txt")):
        shutil.copy(os.path.join(EV, "up_bench.txt"), os.path.join(PR, f"{R}_dense_up_bench.txt"))
    for src, dst in (("prof_overlap", "sp_mnn_b32_kernel_stats.csv"), ("prof_single", "sp_mnn_b32_kernel_stats_single_stream.csv"),
                     ("prof_lg", "sp_lg_b64_kernel_stats.csv"), ("prof_dense", "dense_kernel_stats.csv"),
                     ("prof_b1_mnn", "sp_mnn_b1_kernel_stats.csv"), ("prof_b1_lg", "sp_lg_b1_kernel_stats.csv"),
                     ("prof_events", "events_kernel_stats.csv")):
        ff = glob.glob(os.path.join(EV, src, "**", "*kernel_stats.csv"), recursive=True)
        if not ff:
            continue
        f = ff[0]
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(PR, f"{R}_{dst}"), "w", newline="") as fo:
            w = csv.writer(fo)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
        top = rows[0]
        print(dst, short(top["Name"])[:70], "avg", float(top["AverageNs"]) / 1e3, "us x", top["Calls"])

    def counters(sub):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        f = glob.glob(os.path.join(EV, sub, "**", "*counter_collection.csv"), recursive=True)[0]
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        return agg

    out = {}
    fetch, write = counters("pmc_FETCH_SIZE"), counters("pmc_WRITE_SIZE")
    k1b = [k for k in fetch if "conv_block_kernel<3, 8, 32, 2, 4, 1, 2, 8, true" in k][0]
    k1a = [k for k in fetch if "conv_block_kernel<3, 8, 32, 2, 4, 1, 2, 2, false" in k][0]
    fk, wk = fetch[k1b]["FETCH_SIZE"], write[k1b]["WRITE_SIZE"]
    out["kernel"] = "conv_block_kernel<3,8,32,2,4,1,2,8,true> (3 per CU) (conv1b 64->64 @264x352, B=32)"
    out["FETCH_SIZE_KB"] = sum(fk) / len(fk)
    out["WRITE_SIZE_KB"] = sum(wk) / len(wk)
    out["launches_averaged"] = len(fk)
    # gfx950: FETCH_SIZE tallies a 128-byte request as 64 bytes (MI355X_MICROARCH.md, HBM section), so whether the raw counter or
    # twice it is the byte count depends on the request width the access pattern produces.  Calibrate on conv1a, whose input
    # size is known and which reads with the same 4-byte-per-lane loads in the same tile order.
    c_fetch = sum(fetch[k1a]["FETCH_SIZE"]) / len(fetch[k1a]["FETCH_SIZE"]) * 1024
    c_in = 32 * 260 * 346 * 4
    corr = 2.0 if c_fetch < 0.75 * c_in else 1.0
    out["fetch_correction"] = corr
    out["hbm_bytes_per_launch"] = (corr * out["FETCH_SIZE_KB"] + out["WRITE_SIZE_KB"]) * 1024
    out["algorithmic_bytes_per_launch"] = 32 * (64 * 264 * 352 + 64 * 132 * 176) * 4
    out["note"] = ("separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over `bench.py --kernel-only`. WRITE_SIZE equals the output "
                   f"tensor (190.3 MB). FETCH_SIZE correction x{corr:g}: calibrated on conv1a (known {c_in / 1e6:.1f} MB input, raw counter "
                   f"{c_fetch / 1e6:.1f} MB): with the XCD-contiguous tile order neighbouring lanes' 4-byte loads reach the memory side as "
                   "128-byte requests, which the counter tallies at 64 bytes (the guide's 1/2 rule); before round 2's remap the requests "
                   "were 64 bytes wide and the raw counter was the byte count (r01 file). hbm_bytes_per_launch = correction x FETCH_SIZE + "
                   "WRITE_SIZE.")
    out["conv1a_calibration"] = {"FETCH_SIZE_KB": sum(fetch[k1a]["FETCH_SIZE"]) / len(fetch[k1a]["FETCH_SIZE"]),
                                 "WRITE_SIZE_KB": sum(write[k1a]["WRITE_SIZE"]) / len(write[k1a]["WRITE_SIZE"]),
                                 "input_bytes": 32 * 260 * 346 * 4, "output_bytes": 32 * 64 * 264 * 352 * 4}
    busy = {}
    for sub in ("pmc_busy", "pmc_busy_lg"):
        for k, v in counters(sub).items():
            if "SQ_VALU_MFMA_BUSY_CYCLES" not in v or sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) == 0:
                continue
            b = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / (sum(v["GRBM_GUI_ACTIVE"]) / 8 * 1024)
            busy[k.split("(")[0].replace("void ", "")] = {"launches": len(v["GRBM_GUI_ACTIVE"]), "mfma_busy_frac": round(b, 4)}
    out["mfma_busy_fraction_per_kernel (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs))"] = busy
    json.dump(out, open(os.path.join(PR, f"{R}_pmc_conv1b.json"), "w"), indent=1)
    # round 6: the image extractor's fused first two layers (conv1ab_kernel) from the same passes
    kf = [k for k in fetch if "conv1ab_kernel<1, true" in k]
    if kf:
        ff, wf = fetch[kf[0]]["FETCH_SIZE"], write[kf[0]]["WRITE_SIZE"]
        fo = {"kernel": "conv1ab_kernel<1, true, 6> (image extractor, layers 1-2 as one launch: 1->64->64 3x3 @264x352 + ReLU + pool, B=32)",
              "FETCH_SIZE_KB": sum(ff) / len(ff), "WRITE_SIZE_KB": sum(wf) / len(wf), "launches_averaged": len(ff), "fetch_correction": corr}
        fo["hbm_bytes_per_launch"] = (corr * fo["FETCH_SIZE_KB"] + fo["WRITE_SIZE_KB"]) * 1024
        fo["algorithmic_bytes_per_launch"] = 32 * (260 * 346 + 64 * 132 * 176) * 4
        fo["two_launches_it_replaces_bytes"] = out["conv1a_calibration"]["input_bytes"] + 2 * out["conv1a_calibration"]["output_bytes"] + 32 * 64 * 132 * 176 * 4
        fo["note"] = ("same separate --pmc passes and FETCH_SIZE correction as the conv1b file; the raw image is read through the workgroups' 12x36 "
                      "halo tiles (re-read by neighbouring tiles out of L2), the first layer's 761 MB output is neither written nor read")
        for k, v in busy.items():
            if k.startswith("conv1ab_kernel"):
                fo["mfma_busy"] = v
        json.dump(fo, open(os.path.join(PR, f"{R}_pmc_conv1ab.json"), "w"), indent=1)
        print("pmc fused: hbm bytes/launch", fo["hbm_bytes_per_launch"] / 1e6, "MB")
    write_readme(out, busy)
    print("pmc: hbm bytes/launch", out["hbm_bytes_per_launch"] / 1e6, "MB; busy", {k[:40]: v["mfma_busy_frac"] for k, v in busy.items()})


def write_readme(pmc, busy):
    def bench(name):
        return json.load(open(os.path.join(PR, f"{R}_bench_{name}.json")))

    def stats(name):
        return list(csv.DictReader(open(os.path.join(PR, f"{R}_{name}.csv"))))

    b = bench("sp_mnn_b32")
    rf = b["roofline"]
    ko = None
    f = glob.glob(os.path.join(EV, "prof_kernel_only", "**", "*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(f[0], os.path.join(PR, f"{R}_conv1b_kernel_only_stats.csv"))
        for r in csv.DictReader(open(f[0])):
            if "conv_block_kernel<3, 8, 32, 2, 4, 1, 2, 8, true" in r["Name"]:
                ko = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
    single = [r for r in stats("sp_mnn_b32_kernel_stats_single_stream") if "conv_block_kernel<3, 8, 32, 2, 4, 1, 2, 8, true" in r["Name"]][0]
    over = [r for r in stats("sp_mnn_b32_kernel_stats") if "conv_block_kernel<3, 8, 32, 2, 4, 1, 2, 8, true" in r["Name"]][0]
    flop = rf["flop_per_launch"]
    lines = []
    A = lines.append
    A("# profiles/")
    A("")
    A("Evidence measured on a 1-GPU MI355X box (`gpurun`), named per round.  Collected by `tools/collect_profiles.sh` (on the box) and")
    A("assembled by `tools/assemble_profiles.py` (this file is generated by it).  Kernel names are shortened (`(anonymous namespace)::` removed).")
    A("")
    A("## Bench lines (`python bench.py ...`, inputs resident in HBM, one host sync per step)")
    A("")
    A("| file | command | pairs/s | ms/step | note |")
    A("|---|---|---|---|---|")
    rows = [("sp_mnn_b32", "`bench.py`", "headline: VGG(event)+SuperPoint(image)+MNN, B=32, sparse outputs"),
            ("sp_mnn_b32_full_dict", "`bench.py --log-assignment --dense`", "reference-complete output dict (92 MB/image dense descriptors + log_assignment)"),
            ("sp_mnn_b32_with_metrics", "`bench.py --with-metrics`", "plus MR/MMA/VDD harness metrics on the device"),
            ("sp_lg_b64", "`bench.py --config sp_lg`", "LightGlue matcher, B=64"),
            ("silk_mnn_b32", "`bench.py --config silk_mnn`", "SiLK-shaped extractors (no pooling, 365 GFLOP/pair), B=32"),
            ("silk_lg_b32", "`bench.py --config silk_lg`", "SiLK-shaped extractors + LightGlue (128-d descriptors through input_proj), B=32")]
    for name, cmd, note in rows:
        if not os.path.exists(os.path.join(PR, f"{R}_bench_{name}.json")):
            continue
        j = bench(name)
        A(f"| `{R}_bench_{name}.json` | {cmd} | {j['value']:.0f} | {j['ms_per_step']:.2f} | {note} |")
    cb = b.get("cpu_baseline")
    if cb:
        A("")
        A(f"CPU baseline of the headline line: {cb['value']} {cb['unit']} ({cb['kind']}, {cb['cores']} host threads; {cb['sample']}).")
    A("")
    if b.get("extra_configs"):
        A("")
        A("`extra_configs` of the headline line (short legs inside the same `python bench.py` run, i.e. what the driver's run also measures):")
        A("")
        A("| config | pairs/step | pairs/s | ms/step | mean matches | note |")
        A("|---|---|---|---|---|---|")
        for e in b["extra_configs"]:
            A(f"| {e['config']} | {e['pairs_per_step']} | {e['value']:.0f} | {e['ms_per_step']:.3f} | {e.get('mean_matches')} | {e.get('note', '')[:110]} |")
    if b.get("scale_legs"):
        A("")
        A("`scale_legs` of the headline line (every rank runs them after the headline at EVERY N; at N=1 the SP+LightGlue leg is BASELINE configs[3], "
          "at N=8 it is configs[4] = 512 pairs over 8 GPUs):")
        A("")
        A("| config | pairs per GPU per step | n_gpus | pairs/s (whole job) | ms/step | per-rank min / max pairs/s | steps |")
        A("|---|---|---|---|---|---|---|")
        for e in b["scale_legs"]:
            pr = e["per_rank_pairs_per_s"]
            A(f"| {e['config']} | {e['pairs_per_gpu_per_step']} | {e['n_gpus']} | {e['value']:.0f} | {e['ms_per_step']:.2f} | {pr['min']:.0f} / {pr['max']:.0f} | {e['steps']} |")
    ct = b.get("cpu_baseline_torch")
    if ct:
        A("")
        A(f"`cpu_baseline_torch`: {ct['value']} {ct['unit']} ({ct['kind']}, {ct['threads']} threads; {ct['sample']}).")
    if cb and "verified_pairs" in cb:
        A("")
        A(f"Post-run verification (outside the timed region): {cb['verified_pairs']} of {cb['verified_of']} pairs of the CPU sample have GPU outputs "
          f"equal to the oracle's ({cb['verified_what']}).  `mean_matches` of the headline line: {b['config']['mean_matches']} "
          f"({b['config']['weights']}).")
    for name, tag in (("rccl_one_rank_launcher.log", "bench.py's own launcher"), ("rccl_one_rank_torchrun.log", "torchrun")):
        f = os.path.join(PR, f"{R}_{name}")
        if os.path.exists(f):
            j = last_json_line(f)
            A("")
            A(f"One-rank RCCL launch through {tag} (`{R}_{name}`): {j['value']:.0f} pairs/s, `rccl` = {json.dumps(j.get('rccl'))}.")
    if b.get("roofline_stages"):
        A("")
        A("## Stage rooflines (`roofline_stages` of the headline line: HIP events around every launch of one forward, library-side `einx_profile_*`)")
        A("")
        A("| stage | ms | achieved | fraction of the fp32-MFMA peak | other |")
        A("|---|---|---|---|---|")
        for st in b["roofline_stages"]:
            if st.get("bound") == "hbm":
                A(f"| {st['stage']} | {st['ms']} | {st['achieved']} {st['unit']} | {st['frac'] * 100:.1f} % of 8 TB/s | {json.dumps(st.get('kernels_ms'))} |")
            elif "achieved" in st:
                other = ""
                if "hbm" in st:
                    other = f"HBM view: {st['hbm']['achieved_GBps']:.0f} GB/s algorithmic = {st['hbm']['frac'] * 100:.1f} % of 8 TB/s"
                if "other_kernels_ms" in st:
                    other = "other kernels (ms): " + json.dumps(st["other_kernels_ms"])
                A(f"| {st['stage']} | {st['ms']} | {st['achieved']} {st['unit']} | {st['frac'] * 100:.1f} % | {other} |")
            else:
                A(f"| {st['stage']} | | | | {json.dumps(st.get('kernels_ms'))} |")
    A("")
    A("## Dominant kernel: `conv_block_kernel<3,8,32,2,4,1,2,8,true>` (three workgroups per CU; conv1b, 64->64 3x3 @264x352 + ReLU + 2x2 max-pool, B=32)")
    A("")
    A(f"* algorithmic work per launch: {flop / 1e9:.1f} GFLOP (2 x 64 x 64 x 9 x 264 x 352 x 32); algorithmic bytes {pmc['algorithmic_bytes_per_launch'] / 1e6:.1f} MB.")
    A(f"* `roofline` in the bench line (mean of {rf.get('launches_timed', 10)} per-launch HIP-event pairs on the launch stream): {rf['launch_ms']:.3f} ms -> "
      f"**{rf['achieved']:.1f} TFLOP/s = {rf['frac'] * 100:.1f} %** of the 157.3 TFLOP/s dense fp32-MFMA peak."
      + (f"  Back to back (one event pair around all launches, launch i+1 fills the CUs while launch i drains): {rf['back_to_back_ms']:.3f} ms = "
         f"{rf['back_to_back_TFLOPs']:.1f} TFLOP/s." if "back_to_back_ms" in rf else ""))
    if ko:
        A(f"* `{R}_conv1b_kernel_only_stats.csv` (`rocprofv3 --kernel-trace --stats -- python3 bench.py --kernel-only`, the same launches alone): "
          f"{ko[0]} launches, average {ko[1]:.3f} ms = {flop / ko[1] / 1e9:.1f} TFLOP/s.")
    A(f"* `{R}_sp_mnn_b32_kernel_stats_single_stream.csv` (`EINX_OVERLAP=0 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 "
      f"--no-cpu-baseline`): {single['Calls']} launches, average {float(single['AverageNs']) / 1e6:.3f} ms = {flop / float(single['AverageNs']) * 1e9 / 1e12:.1f} TFLOP/s. "
      "The average mixes the roofline-loop launches and the image-side layer with the event-side layer of the same shape (dense, not ReLU-sparse, "
      "inputs: ~2 % slower) and the first launches of the process, which run before the device reaches its working clocks.")
    A(f"* `{R}_sp_mnn_b32_kernel_stats.csv` (same command, default two-stream schedule): average {float(over['AverageNs']) / 1e6:.3f} ms -- "
      "the event and image extractors run concurrently, so per-kernel durations there include time-sharing of the CUs; use the single-stream file for kernel rates.")
    A(f"* `{R}_pmc_conv1b.json`: separate `--pmc` passes over `bench.py --kernel-only`: {pmc.get('fetch_correction', 1.0):g} x FETCH_SIZE "
      f"{pmc['FETCH_SIZE_KB'] / 1024:.0f} MiB + WRITE_SIZE {pmc['WRITE_SIZE_KB'] / 1024:.0f} MiB = {pmc['hbm_bytes_per_launch'] / 1e6:.0f} MB per launch "
      f"(this is `roofline.traffic`) vs {pmc['algorithmic_bytes_per_launch'] / 1e6:.0f} MB algorithmic.  The correction factor is calibrated on "
      "conv1a's known input size (gfx950 tallies 128-byte requests at 64 bytes; see the json's note).  Round 1 (round-robin tile order): "
      "1537 MB per launch, i.e. the halo re-reads and cross-XCD duplicates that the XCD-contiguous order removed.")
    A("* MFMA-busy (`SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs)`, from the same json):")
    A("")
    A("| kernel | launches | MFMA busy |")
    A("|---|---|---|")
    for k, v in sorted(busy.items(), key=lambda kv: -kv[1]["mfma_busy_frac"]):
        A(f"| `{k}` | {v['launches']} | {v['mfma_busy_frac'] * 100:.1f} % |")
    fz = rf.get("fused_first_two_layers")
    if fz:
        A("")
        A("## Fused first two layers of the image extractor: `conv1ab_kernel<1, true, 6>` (round 6)")
        A("")
        A(f"* one launch instead of two: {fz['launch_ms']:.3f} ms per launch (mean of {fz['launches_timed']} per-launch HIP-event pairs) against "
          f"{fz['replaces_ms']['first_layer']:.3f} + {fz['replaces_ms']['second_layer']:.3f} = {fz['replaces_ms']['sum']:.3f} ms for the two launches alone; "
          f"{fz['flop_per_launch'] / 1e9:.1f} GFLOP (both layers) -> {fz['achieved']:.1f} TFLOP/s = {fz['frac'] * 100:.1f} % of the fp32-MFMA peak.")
        fk = os.path.join(EV, "prof_kernel_only")
        for f_ in glob.glob(os.path.join(fk, "**", "*kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f_)):
                if "conv1ab_kernel<1, true" in r["Name"]:
                    A(f"* `{R}_conv1b_kernel_only_stats.csv` lists the same launches: {r['Calls']} launches, average {float(r['AverageNs']) / 1e6:.3f} ms.")
        pj = os.path.join(PR, f"{R}_pmc_conv1ab.json")
        if os.path.exists(pj):
            pf = json.load(open(pj))
            A(f"* `{R}_pmc_conv1ab.json`: {pf['fetch_correction']:g} x FETCH_SIZE {pf['FETCH_SIZE_KB'] / 1024:.0f} MiB + WRITE_SIZE {pf['WRITE_SIZE_KB'] / 1024:.0f} MiB = "
              f"{pf['hbm_bytes_per_launch'] / 1e6:.0f} MB per launch against {pf['algorithmic_bytes_per_launch'] / 1e6:.0f} MB algorithmic (raw image in, pooled "
              f"second-layer output out); the two launches it replaces moved {pf['two_launches_it_replaces_bytes'] / 1e6:.0f} MB.")
    A("")
    A("## Per-kernel time (single stream, exclusive timings), SP+MNN B=32, 6 steps")
    A("")
    A("| kernel | calls | avg us | % |")
    A("|---|---|---|---|")
    for r in stats("sp_mnn_b32_kernel_stats_single_stream")[:16]:
        A(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |")
    A("")
    A(f"## LightGlue configuration (`{R}_sp_lg_b64_kernel_stats.csv`, `rocprofv3 --kernel-trace --stats -- python3 bench.py --config sp_lg --steps 2 --warmup 1`)")
    A("")
    A("| kernel | calls | avg us | % |")
    A("|---|---|---|---|")
    for r in stats("sp_lg_b64_kernel_stats")[:12]:
        A(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |")
    A("")
    dk = os.path.join(PR, f"{R}_dense_kernel_stats.csv")
    if os.path.exists(dk):
        A(f"## Dense descriptor maps alone (`{R}_dense_kernel_stats.csv`: `rocprofv3 --kernel-trace --stats -- python3 tools/up_bench.py`, B=32, one side, 2.95 GB of output)")
        A("")
        A("| kernel | calls | avg us |")
        A("|---|---|---|")
        for r in csv.DictReader(open(dk)):
            if "upsample" in r["Name"]:
                A(f"| `{r['Name'][:80]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} |")
        ub = os.path.join(PR, f"{R}_dense_up_bench.txt")
        if os.path.exists(ub):
            A("")
            A("`tools/up_bench.py --ref` (HIP events): " + " / ".join(ln.strip() for ln in open(ub).read().splitlines() if ln.strip()))
        A("")
    for tag, title in (("sp_mnn_b1", "SP+MNN"), ("sp_lg_b1", "SP+LightGlue")):
        bk = os.path.join(PR, f"{R}_{tag}_kernel_stats.csv")
        if not os.path.exists(bk):
            continue
        rows = list(csv.DictReader(open(bk)))
        nf = 620.0  # forwards of tools/latency_b1.py: 20 warm-up + 3 x 200
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        A(f"## Single pair, {title} (`{R}_{tag}_kernel_stats.csv`: `rocprofv3 --kernel-trace --stats -- python3 tools/latency_b1.py 1 {'SP_MNN' if 'mnn' in tag else 'SP_LG'}`, 620 forwards)")
        A("")
        A(f"Kernel time per forward (sum over both sides / all streams): {tot / nf / 1e3:.0f} us.")
        A("")
        A("| kernel | launches per forward | avg us | us per forward |")
        A("|---|---|---|---|")
        for r in rows[:14]:
            A(f"| `{short(r['Name'])[:80]}` | {int(r['Calls']) / nf:.1f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / nf / 1e3:.0f} |")
        lt = os.path.join(PR, f"{R}_latency_b1{'_sp_lg' if 'lg' in tag else ''}.txt")
        if os.path.exists(lt):
            A("")
            A("`tools/latency_b1.py` without the profiler: " + open(lt).read().strip().splitlines()[-1])
        A("")
    tp = os.path.join(PR, f"{R}_pmc_conv_tiles.json")
    if os.path.exists(tp):
        A(f"## Per-variant conv counters (`{R}_pmc_conv_tiles.json`, `tools/experiments/r3_pmc_conv.sh` + `tools/pmc_tiles.py`)")
        A("")
        tj = json.load(open(tp))
        if "kernels" in tj:  # round 5: one build, per kernel variant
            A(tj.get("what", ""))
            A("")
            A(tj.get("reading", ""))
            A("")
            A("| kernel variant | MFMA busy | VALU / MFMA (incl. the MFMAs) | LDS / MFMA | effective MHz | mean us (profiled) |")
            A("|---|---|---|---|---|---|")
            for k, v in tj["kernels"].items():
                A(f"| `{k}` | {v.get('mfma_busy_frac')} | {v.get('valu_per_mfma')} | {v.get('lds_per_mfma')} | {v.get('effective_mhz'):.0f} | {v.get('mean_us_per_launch_profiled')} |")
            A("")
            tj = {"conflict_free_pitch": {}, "plain_pitch": {}}
        A("| kernel variant | LDS conflicts / idx-active (plain pitch -> shipped) | MFMA busy (plain -> shipped) | effective MHz | mean us (profiled) |") if tj["conflict_free_pitch"] else None
        A("|---|---|---|---|---|") if tj["conflict_free_pitch"] else None
        for k, v in tj["conflict_free_pitch"].items():
            o = tj["plain_pitch"].get(k, {})
            A(f"| `{k}` | {o.get('lds_bank_conflict_per_idx_active')} -> {v.get('lds_bank_conflict_per_idx_active')} | {o.get('mfma_busy_frac')} -> "
              f"{v.get('mfma_busy_frac')} | {v.get('effective_mhz'):.0f} | {o.get('mean_us_per_launch_profiled')} -> {v.get('mean_us_per_launch_profiled')} |")
        A("")
    pe = os.path.join(PR, f"{R}_parity_errors.json")
    if os.path.exists(pe):
        A(f"## Measured float errors of the tolerance-based parity tests (`{R}_parity_errors.json`, written by `pytest -m gpu`)")
        A("")
        A("| comparison | measured max abs error | tolerance | largest reference magnitude |")
        A("|---|---|---|---|")
        pj = json.load(open(pe))
        for k, v in sorted(pj.items()):
            if k != "assignment_flips":
                A(f"| {k} | {v['max_abs_err']:.3e} | {v['atol']:.1e} | {v['max_abs_ref']:.3g} |")
        A("")
        if "assignment_flips" in pj:
            A("Match-assignment flips per comparison (`helpers.record_flips`; -1 = unmatched counts as an assignment): "
              "the LightGlue log_assignment bounds are multiples of the reference's own noise floor (`tests/golden/lgcal.npz`, notes below).")
            A("")
            A("| comparison | flips | assignments compared | matched in the checker |")
            A("|---|---|---|---|")
            for k, v in sorted(pj["assignment_flips"].items()):
                A(f"| {k} | {v['flips']} | {v['compared']} | {v['matched']} |")
            A("")
    extra = os.path.join(PR, f"{R}_notes.md")
    if os.path.exists(extra):  # hand-written findings of the round (experiments, A/B runs), kept next to the raw logs
        A(open(extra).read())
    older = sorted(f for f in os.listdir(PR) if re.match(r"r\d+_", f) and not f.startswith(R + "_"))
    if older:
        A("## Earlier rounds")
        A("")
        A("Files of earlier rounds are kept as they were: " + ", ".join(f"`{f}`" for f in older) + ".")
        A("")
    open(os.path.join(PR, "README.md"), "w").write("\n".join(lines))


if __name__ == "__main__":
    main()
