#!/bin/bash
# kNN kernel time for several grid granularities (workgroups per CU wanted by the planner)
mkdir -p gpurun_out
for w in 2 4 8 16; do
  VO_KNN_WG_PER_CU=$w timeout -k 10 200 python -u bench.py --steps 10 --warmup 3 --no-cpu > gpurun_out/sweep$w.log 2>&1 || { tail -5 gpurun_out/sweep$w.log; exit 1; }
  python -c "
import json;l=json.loads(open('gpurun_out/sweep$w.log').read().strip().splitlines()[-1]);print('wg_per_cu $w knn2 us', round(l['kernel_avg_us']['knn2'],1))"
done
