#!/bin/bash
# One rocprofv3 --pmc pass (counters given as arguments) over a short bench run; prints per-kernel sums.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -o pmc -- python3 bench.py --inproc --steps 2 --warmup 1 --no-cpu --no-latency > gpurun_out/pmc_$tag.log 2>&1 || { tail -20 gpurun_out/pmc_$tag.log; exit 1; }
f=$(find gpurun_out/pmc_$tag -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    short = "knn2" if "knn2_partial" in k else "score<0>" if ("score_kernel<0>" in k or "score_fund_mfma" in k) else "score<2>" if "score_kernel<2>" in k else "solve_7pt" if "solve_7pt" in k else "solve_p3p" if "solve_p3p" in k else None
    if not short: continue
    acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(short, r["Counter_Name"])] += 1
for k in acc:
    print(k, {c: (round(v / n[(k, c)]), n[(k, c)]) for c, v in acc[k].items()})
PY
