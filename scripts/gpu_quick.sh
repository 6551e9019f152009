#!/bin/bash
# quick GPU check: the given pytest selection, then bench lines.  BENCH_VARIANTS="<args>|<args>" overrides the two default legs
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread "$@" > gpurun_out/quick_pytest.log 2>&1 || { tail -40 gpurun_out/quick_pytest.log; exit 1; }
tail -2 gpurun_out/quick_pytest.log
IFS='|' read -ra LEGS <<< "${BENCH_VARIANTS:-|--no-pose-tail}"
for extra in "${LEGS[@]}"; do
timeout -k 10 300 python3 bench.py --inproc --steps 10 --warmup 3 --no-cpu --no-latency $extra > gpurun_out/quick_bench.log 2>&1 || { tail -20 gpurun_out/quick_bench.log; exit 1; }
tail -1 gpurun_out/quick_bench.log | python3 -c "
import json,sys; l=json.loads(sys.stdin.read())
print('$extra', 'ms/step', round(l['ms_per_step'],3), {k: round(v) for k, v in l['kernel_avg_us'].items()}, 'roofline', l['roofline']['kernel'], l['roofline'].get('frac'), [(r['kernel'], r.get('frac') and round(r['frac'],3)) for r in l['roofline_other']], l['ransac_scoring']['final_niters_first_pairs'], l['ransac_scoring'].get('two_view_pairs_needing_more_samples'))"
done
