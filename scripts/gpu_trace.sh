#!/bin/bash
# rocprofv3 kernel trace of a short bench run; prints the top kernels.  usage: scripts/gpu_trace.sh <tag> [bench args]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
tag=$1; shift
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o bench -- python3 bench.py --inproc --no-cpu --no-latency --steps 10 --warmup 2 "$@" > gpurun_out/prof_$tag.log 2>&1 || { tail -30 gpurun_out/prof_$tag.log; exit 1; }
python3 - gpurun_out/prof_$tag/bench_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f} {r['Percentage']}%")
PY
