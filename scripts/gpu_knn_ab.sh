#!/bin/bash
# A/B of kNN sweep kernel builds through the bench: in-tree library first, then every ablate/libvo_knn_*.so
# usage: scripts/gpu_knn_ab.sh [test] [cfg ...]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out
if [ "$1" = test ]; then shift
timeout -k 10 600 python -u -m pytest tests/test_knn_gpu.py tests/test_golden_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread -k "knn or match" > gpurun_out/knn_ab_pytest.log 2>&1 || { tail -30 gpurun_out/knn_ab_pytest.log; exit 1; }
tail -2 gpurun_out/knn_ab_pytest.log
fi
for lib in "" $(ls ablate/libvo_knn_*.so 2>/dev/null); do
  for cfg in ${@:-cfg4 cfg5}; do
    VO_HIP_LIB=${lib:+$PWD/$lib} timeout -k 10 300 python -u bench.py --inproc --config $cfg --steps 8 --warmup 2 --no-cpu --no-latency > gpurun_out/knn_ab.log 2>&1 || { tail -5 gpurun_out/knn_ab.log; exit 1; }
    python3 -c "
import json;l=json.loads(open('gpurun_out/knn_ab.log').read().strip().splitlines()[-1]);r=[x for x in [l['roofline']]+l['roofline_other'] if x['kernel'].startswith('knn2')][0]
print('${lib:-in-tree}', '$cfg', 'knn2 us', round(l['kernel_avg_us']['knn2'],1), 'frac', round(r['frac'],3), 'ms/step', round(l['ms_per_step'],3))"
  done
done
