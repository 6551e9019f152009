#!/bin/bash
# SQ co-execution / wait counters of the kNN sweep kernel (and clock under load = GRBM_GUI_ACTIVE / 8 / duration) over a
# short bench run; two separate --pmc passes, the program itself after `--`.  usage: scripts/gpu_pmc_knn.sh <tag> [bench args]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
tag=${1:-r3}; shift
mkdir -p gpurun_out
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  p=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/${tag}_pmc_$p -o pmc -- python3 bench.py --inproc --steps 2 --warmup 1 --no-cpu --no-latency "$@" > gpurun_out/${tag}_pmc_$p.log 2>&1 || { tail -5 gpurun_out/${tag}_pmc_$p.log; exit 1; }
done
python3 - $tag <<'PY'
import csv, sys, json, collections, glob
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/{tag}_pmc_*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        s = "knn2_partial_kernel" if "knn2_partial" in k else "score_kernel<F>" if "score_kernel<0>" in k else "screen_fund_mfma_kernel" if "screen_fund_mfma" in k else "count_contenders_kernel" if "count_contenders" in k else None
        if s:
            acc[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "SQ_INSTS_LDS"):
                acc[s]["duration_us:" + r["Counter_Name"]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
out = {}
for s, d in acc.items():
    o = {c: sum(v) / len(v) for c, v in d.items()}
    dur = o.get("duration_us:GRBM_GUI_ACTIVE")
    if dur and "GRBM_GUI_ACTIVE" in o:
        o["clock_ghz"] = o["GRBM_GUI_ACTIVE"] / 8 / dur / 1e3
    if "SQ_VALU_MFMA_BUSY_CYCLES" in o and o["SQ_VALU_MFMA_BUSY_CYCLES"]:
        o["coexec_share_of_mfma_busy"] = o["SQ_VALU_MFMA_COEXEC_CYCLES"] / o["SQ_VALU_MFMA_BUSY_CYCLES"]
    if "SQ_WAVE_CYCLES" in o:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            o[c + "_share_of_wave_cycles"] = o[c] / o["SQ_WAVE_CYCLES"]
    out[s] = o
json.dump(out, open(f"gpurun_out/{tag}_knn_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
