#!/bin/bash
# A/B of whole-library builds through the bench: in-tree first, then every library matching the glob in $1 (two runs each)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
pat=$1; shift
for lib in "" $(ls $pat 2>/dev/null); do
  for rep in 1 2; do
  VO_HIP_LIB=${lib:+$PWD/$lib} timeout -k 10 300 python3 bench.py --inproc --steps ${AB_STEPS:-20} --warmup 3 --no-cpu --no-latency "$@" > gpurun_out/ab_libs.log 2>&1 || { tail -20 gpurun_out/ab_libs.log; exit 1; }
  tail -1 gpurun_out/ab_libs.log | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('${lib:-in-tree}', 'ms/step', round(l['ms_per_step'],3), {k: round(v) for k, v in l['kernel_avg_us'].items()})"
  done
done
