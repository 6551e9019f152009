#!/bin/bash
# A/B of bench.py under environment settings: each argument is "NAME=VALUE[,NAME=VALUE...]" (or "-" for none)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for setting in "$@"; do
  envs=$(echo "$setting" | tr ',' ' '); [ "$setting" = "-" ] && envs=""
  for rep in 1 2; do
  env $envs timeout -k 10 300 python3 bench.py --inproc --steps 20 --warmup 3 --no-cpu --no-latency $BENCH_ARGS > gpurun_out/ab_env.log 2>&1 || { tail -20 gpurun_out/ab_env.log; exit 1; }
  tail -1 gpurun_out/ab_env.log | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('$setting', 'ms/step', round(l['ms_per_step'],3), {k: round(v) for k, v in l['kernel_avg_us'].items()})"
  done
done
