#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command (GPU box).  Output under gpurun_out/prof_*.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=${1:-r1}; shift
mkdir -p gpurun_out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o bench -- python3 bench.py --inproc --steps 20 --warmup 3 --no-cpu --no-latency "$@" > gpurun_out/prof_$tag.log 2>&1 || { tail -30 gpurun_out/prof_$tag.log; exit 1; }
tail -2 gpurun_out/prof_$tag.log
find gpurun_out/prof_$tag -name '*stats*' | head
f=$(find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then head -30 "$f"; fi
exit 0
