#!/bin/bash
# A/B of library builds on one box: scripts/diag/ab_libs.sh lib1.so lib2.so ...  (each in a fresh process, bench without CPU legs)
for lib in "$@"; do
  VO_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu --no-latency > gpurun_out/ab_$(basename $lib .so).log 2>&1 || { tail -5 gpurun_out/ab_$(basename $lib .so).log; exit 1; }
  python - "$lib" <<'PY'
import json, sys, os
lib = sys.argv[1]
d = json.loads(open("gpurun_out/ab_%s.log" % os.path.basename(lib)[:-3]).read().strip().splitlines()[-1])
print(lib, round(d["value"]), round(d["ms_per_step"], 3), {k: round(v) for k, v in d.get("kernel_avg_us", {}).items()})
PY
done
