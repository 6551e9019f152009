#!/bin/bash
# A/B timing of prebuilt library variants (GPU box): bash scripts/lib_variants.sh <rounds> <steps> lib1.so lib2.so ...
# ("base" = the in-tree library).  Variants are interleaved round by round so that drift hits all alike.
cd "$(dirname "$0")/.."
rounds=$1; steps=$2; shift 2
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    p=""; [ "$lib" != base ] && p=$PWD/$lib
    VO_HIP_LIB=$p timeout -k 10 300 python bench.py --steps $steps --warmup 3 --no-cpu > gpurun_out/lv.log 2>&1 || { tail -5 gpurun_out/lv.log; exit 1; }
    python -c "
import json;l=json.loads(open('gpurun_out/lv.log').read().strip().splitlines()[-1]);print('round $r', '$lib', round(l['ms_per_step'],3))"
  done
done
