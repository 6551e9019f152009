#!/usr/bin/env python3
"""gpurun_out/<tag>_counters.json from the rocprofv3 outputs of scripts/gpu_roofline.sh: per kernel the average
duration (kernel trace), SQ counters + clock under load (GRBM_GUI_ACTIVE / 8 / duration of the same dispatch),
HBM bytes per launch (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md).  Copy to profiles/<tag>_counters.json."""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1]
bench_args = sys.argv[2] if len(sys.argv) > 2 else ""
SHORT = (("knn2_partial", "knn2_partial_kernel"), ("score_kernel<0>", "score_fund_kernel"), ("score_fund_mfma_kernel", "score_fund_kernel"), ("screen_fund_mfma", "screen_fund_kernel"), ("count_contenders", "count_contenders_kernel"), ("score_kernel<1>", "score_ess_kernel"),
         ("score_kernel<2>", "score_pnp_kernel"), ("solve_7pt", "solve_7pt_kernel"), ("solve_p3p", "solve_p3p_kernel"),
         ("desc_f32_to_i8", "desc_f32_to_i8_kernel"), ("gen_subsets", "gen_subsets_kernel"), ("epnp_batch", "epnp_batch_kernel"))


def short(name):
    for needle, s in SHORT:
        if needle in name:
            return s
    return None


out = collections.defaultdict(dict)
# kernel trace: average duration per kernel over the 20 timed steps + warm-up
for f in glob.glob(f"gpurun_out/prof_{tag}/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        s = short(r["Name"])
        if s:
            out[s]["trace_calls"] = out[s].get("trace_calls", 0) + int(r["Calls"])
            out[s]["trace_total_ns"] = out[s].get("trace_total_ns", 0) + int(float(r["TotalDurationNs"]))
for s in out:
    out[s]["trace_avg_us"] = out[s]["trace_total_ns"] / out[s]["trace_calls"] / 1e3


def pmc(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmc_{tag}_{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            s = short(r["Kernel_Name"])
            if s:
                acc[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
                acc[s]["_dur_ns_" + r["Counter_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return acc


sq = pmc("sq")
for s, d in sq.items():
    # the largest dispatches only (the F scoring kernel also runs tiny warm-up shaped launches in other stages)
    for c, v in d.items():
        if not c.startswith("_"):
            out[s][c] = sum(v) / len(v)
    if "GRBM_GUI_ACTIVE" in d:
        dur = d["_dur_ns_GRBM_GUI_ACTIVE"]
        out[s]["pmc_pass_avg_us"] = sum(dur) / len(dur) / 1e3
        out[s]["clock_ghz"] = sum(d["GRBM_GUI_ACTIVE"]) / 8.0 / sum(dur)
for sub, cn in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    for s, d in pmc(sub).items():
        if cn in d:
            out[s][cn + "_KB_per_launch"] = sum(d[cn]) / len(d[cn])
for s, d in out.items():
    if "FETCH_SIZE_KB_per_launch" in d and "WRITE_SIZE_KB_per_launch" in d:
        d["hbm_bytes_per_launch"] = int(2 * d["FETCH_SIZE_KB_per_launch"] * 1024 + d["WRITE_SIZE_KB_per_launch"] * 1024)
try:
    line = json.loads(open(f"gpurun_out/{tag}_bench_line.json").read())
    pairs = line["config"]["pairs_per_gpu_per_step"]
    import re
    m = re.search(r"(\d+) \(model, correspondence\) evaluations", json.dumps(line))
    if m and "score_fund_kernel" in out:
        out["score_fund_kernel"]["evaluations_per_launch"] = int(m.group(1))
except Exception as e:                                    # noqa: BLE001
    pairs = None
res = dict(out)
res["_pairs"] = pairs
res["_config"] = "cfg5" if "cfg5" in bench_args else "cfg4"
res["_ransac"] = "ess" if "ess" in bench_args else "fund"
res["_how"] = ("scripts/gpu_roofline.sh: `python3 bench.py --inproc --no-cpu --no-latency " + bench_args + "` under rocprofv3 "
               "--kernel-trace --stats (20 steps; trace_avg_us) and three separate --pmc passes of 2 steps each: SQ_* + GRBM_GUI_ACTIVE, "
               "FETCH_SIZE, WRITE_SIZE, averaged over the kernel's dispatches.  clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / dispatch "
               "duration of the same pass; hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 tallies 128-B reads at 64 B); "
               "SQ_INSTS_VALU counts wave-instructions (x64 lanes = lane-instructions); SQ_*_CYCLES in quad-cycles")
print(json.dumps(res, indent=1))
