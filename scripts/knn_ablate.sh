#!/bin/bash
# Timing experiments on the kNN sweep kernel: builds of libvo_hip.so with parts of the inner loop
# removed (1: no top-2 epilogue, 2: no MFMA, 4: no staging, 5: no staging and no barriers) (results are wrong by construction; only the kernel time is read).  Build here (CPU box):
#   bash scripts/knn_ablate.sh build      then on the GPU box:   bash scripts/knn_ablate.sh run
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  # the ablation branches (wrong results by construction) are NOT in the shipped sources: they live in
  # scripts/experiments/ablation_branches.patch (made against the round-3 tree) and are applied to a scratch copy
  rm -rf ablate/src && mkdir -p ablate/src && cp visual_odometry_amd/csrc/*.h visual_odometry_amd/csrc/*.hip ablate/src/
  patch -d ablate/src -p0 < scripts/experiments/ablation_branches.patch
  for v in 1 2 4 5; do
    d=ablate/obj$v; mkdir -p $d
    for f in vo_runtime vo_knn vo_score vo_ransac vo_solvers vo_epnp vo_epnp_dev vo_pipeline vo_pose; do
      if [ $f = vo_knn ] || [ ! -f $d/$f.o ]; then
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DVO_ABLATE=$v -Iinclude -Ivisual_odometry_amd/csrc -c ablate/src/$f.hip -o $d/$f.o
      fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $d/*.o -o ablate/libvo_hip_abl$v.so
  done
else
  for v in 0 1 2 4 5; do
    lib=""; [ $v != 0 ] && lib=$PWD/ablate/libvo_hip_abl$v.so
    VO_HIP_LIB=$lib timeout -k 10 200 python -u bench.py --steps 10 --warmup 3 --no-cpu > gpurun_out/abl$v.log 2>&1 || { tail -5 gpurun_out/abl$v.log; exit 1; }
    python -c "
import json;l=json.loads(open('gpurun_out/abl$v.log').read().strip().splitlines()[-1]);print('ablate $v knn2 us', round(l['kernel_avg_us']['knn2'],1))"
  done
fi
