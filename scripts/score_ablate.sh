#!/bin/bash
# Timing experiment on the F scoring kernel: a build that never takes the exact path after the
# filter (results wrong near the threshold by construction; only the kernel time is read).
#   bash scripts/score_ablate.sh build   (CPU box)      bash scripts/score_ablate.sh run   (GPU box)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  # the ablation branches (wrong results by construction) are NOT in the shipped sources: they live in
  # scripts/experiments/ablation_branches.patch (made against the round-3 tree) and are applied to a scratch copy
  rm -rf ablate/src && mkdir -p ablate/src && cp visual_odometry_amd/csrc/*.h visual_odometry_amd/csrc/*.hip ablate/src/
  patch -d ablate/src -p0 < scripts/experiments/ablation_branches.patch
  for v in 1 2; do
    d=ablate/sobj$v; mkdir -p $d
    for f in vo_runtime vo_knn vo_score vo_ransac vo_solvers vo_epnp vo_epnp_dev vo_pipeline vo_pose; do
      if [ $f = vo_score ] || [ ! -f $d/$f.o ]; then
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w -Iinclude -DVO_SCORE_ABLATE=$v -Iinclude -Ivisual_odometry_amd/csrc -c ablate/src/$f.hip -o $d/$f.o
      fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $d/*.o -o ablate/libvo_hip_sabl$v.so
  done
else
  for v in 0 1; do
    lib=""; [ $v != 0 ] && lib=$PWD/ablate/libvo_hip_sabl$v.so
    VO_HIP_LIB=$lib timeout -k 10 200 python -u bench.py --steps 10 --warmup 3 --no-cpu > gpurun_out/sabl$v.log 2>&1 || { tail -5 gpurun_out/sabl$v.log; exit 1; }
    python -c "
import json;l=json.loads(open('gpurun_out/sabl$v.log').read().strip().splitlines()[-1]);print('score ablate $v us', l['kernel_avg_us'])"
  done
fi
