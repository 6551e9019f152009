#!/bin/bash
# A/B of the bench step over the number of F iterations counted exactly before pruning starts (0 = no pruning)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for cfg in ${AB_CFGS:-cfg4}; do
for fb in ${AB_FB:-0 64 256 1024 0 256}; do
  VO_BENCH_F_FIRST_BLOCK=$fb timeout -k 10 300 python3 bench.py --inproc --steps ${AB_STEPS:-30} --warmup 3 --no-cpu --no-latency --config $cfg "$@" > gpurun_out/ab_fb.log 2>&1 || { tail -20 gpurun_out/ab_fb.log; exit 1; }
  tail -1 gpurun_out/ab_fb.log | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('$cfg first_block_f=$fb', 'ms/step', round(l['ms_per_step'],3), {k: round(v) for k, v in l['kernel_avg_us'].items()})"
done; done
