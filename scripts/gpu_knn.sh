#!/bin/bash
# quick loop for kNN kernel work: parity tests of the matcher, then the bench (no CPU leg)
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_knn_gpu.py tests/test_golden_gpu.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -4 || exit 1
timeout -k 10 200 python -u bench.py --steps 10 --warmup 3 --no-cpu "$@" > gpurun_out/bench_knn.log 2>&1 || { tail -20 gpurun_out/bench_knn.log; exit 1; }
python - <<'PY'
import json
l=json.loads(open('gpurun_out/bench_knn.log').read().strip().splitlines()[-1])
print("value", round(l["value"],1), "ms/step", round(l["ms_per_step"],3), "avg_us", {k:round(v,1) for k,v in l["kernel_avg_us"].items()})
for r in [l["roofline"]]+l["roofline_other"]: print(r["kernel"], round(r["achieved"],1), r["unit"], "frac", round(r["frac"],3))
PY
