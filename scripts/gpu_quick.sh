#!/bin/bash
# quick GPU check: the given pytest selection, then bench lines (cfg4 with and without the pose tail)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread "$@" > gpurun_out/quick_pytest.log 2>&1 || { tail -40 gpurun_out/quick_pytest.log; exit 1; }
tail -2 gpurun_out/quick_pytest.log
for extra in "" "--no-pose-tail"; do
timeout -k 10 300 python3 bench.py --inproc --steps 10 --warmup 3 --no-cpu --no-latency $extra > gpurun_out/quick_bench.log 2>&1 || { tail -20 gpurun_out/quick_bench.log; exit 1; }
tail -1 gpurun_out/quick_bench.log | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('$extra', 'ms/step', round(l['ms_per_step'],3), {k: round(v) for k, v in l['kernel_avg_us'].items()})"
done
