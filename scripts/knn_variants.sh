#!/bin/bash
# Build (CPU box: `build`) and time (GPU box: `run`) shape variants of the kNN sweep kernel.
# variant = SETS:WAVES:TILES:MINW:CHAINS
set -e
cd "$(dirname "$0")/.."
VARIANTS="2:8:4:2:2 2:8:4:2:4 2:8:4:2:1 2:8:4:3:2 1:8:4:4:4 2:4:4:2:4 2:8:8:2:4 4:4:4:1:2"
if [ "$1" = build ]; then
  mkdir -p ablate/objv
  for f in vo_runtime vo_score vo_ransac vo_solvers vo_epnp vo_epnp_dev vo_pipeline vo_pose; do
    [ -f ablate/objv/$f.o ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w -c visual_odometry_amd/csrc/$f.hip -o ablate/objv/$f.o
  done
  for v in $VARIANTS; do
    IFS=: read S W T M C <<< "$v"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w -DVO_KNN_SETS=$S -DVO_KNN_WAVES=$W -DVO_KNN_TILES=$T -DVO_KNN_MINW=$M -DVO_KNN_CHAINS=$C -Iinclude \
      -Rpass-analysis=kernel-resource-usage -c visual_odometry_amd/csrc/vo_knn.hip -o ablate/objv/knn_$S$W$T$M$C.o 2>&1 | grep -A8 "knn2_partial" | grep -E "VGPRs:|Scratch" | tr '\n' ' '; echo " <- $v"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ablate/objv/vo_*.o ablate/objv/knn_$S$W$T$M$C.o -o ablate/libvo_knn_$S$W$T$M$C.so
  done
else
  mkdir -p gpurun_out
  for v in $VARIANTS; do
    IFS=: read S W T M C <<< "$v"
    VO_HIP_LIB=$PWD/ablate/libvo_knn_$S$W$T$M$C.so timeout -k 10 200 python -u bench.py --steps 10 --warmup 3 --no-cpu > gpurun_out/var.log 2>&1 || { tail -5 gpurun_out/var.log; continue; }
    python -c "
import json;l=json.loads(open('gpurun_out/var.log').read().strip().splitlines()[-1]);print('variant $v knn2 us', round(l['kernel_avg_us']['knn2'],1))"
  done
fi
