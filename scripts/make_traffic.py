#!/usr/bin/env python3
"""profiles/r1_traffic.json from two rocprofv3 --pmc passes (scripts/gpu_pmc.sh f FETCH_SIZE; ... w WRITE_SIZE)."""
import collections, csv, json, sys
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
out = {}
for tag, cn in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f"gpurun_out/pmc_{tag}/pmc_counter_collection.csv")):
        k = r["Kernel_Name"]
        short = ("knn2_partial_kernel" if "knn2_partial" in k else "score_fund_kernel" if "score_kernel<0>" in k
                 else None)
        if short and r["Counter_Name"] == cn:
            acc[short].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        per = sum(v) / len(v)
        out.setdefault(k, {})[cn + "_KB_per_launch"] = per
for k, d in out.items():
    d["hbm_bytes_per_launch"] = int(2 * d["FETCH_SIZE_KB_per_launch"] * 1024 + d["WRITE_SIZE_KB_per_launch"] * 1024)
out["_pairs"] = pairs
out["_how"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of `python3 bench.py --steps 2 --warmup 1 "
               f"--no-cpu --no-latency` ({pairs} frame pairs per launch), averaged over launches; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at "
               "64 B); KB as reported by rocprofv3")
json.dump(out, open("profiles/r1_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
