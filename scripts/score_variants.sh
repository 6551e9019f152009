#!/bin/bash
# Shape sweep of the F scoring kernel (group size x register budget); correct builds, only the
# kernel time differs.   bash scripts/score_variants.sh build   |   bash scripts/score_variants.sh run
set -e
cd "$(dirname "$0")/.."
VARS="${VARS:-4_1 4_6 4_7 4_8}"
if [ "$1" = build ]; then
  mkdir -p ablate/vobj
  for f in vo_runtime vo_knn vo_ransac vo_solvers vo_epnp vo_epnp_dev vo_pipeline vo_pose; do
    [ -f ablate/vobj/$f.o ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w -Iinclude -c visual_odometry_amd/csrc/$f.hip -o ablate/vobj/$f.o
  done
  for v in $VARS; do
    g=${v%_*}; w=${v#*_}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w -Iinclude -DVO_SCORE_G=$g -DVO_SCORE_MINW=$w -c visual_odometry_amd/csrc/vo_score.hip -o ablate/vo_score_$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ablate/vobj/*.o ablate/vo_score_$v.o -o ablate/libvo_score_$v.so
  done
else
  for v in $VARS; do
    VO_HIP_LIB=$PWD/ablate/libvo_score_$v.so timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu --pairs 256 > gpurun_out/sv$v.log 2>&1 || { tail -5 gpurun_out/sv$v.log; exit 1; }
    python -c "
import json;l=json.loads(open('gpurun_out/sv$v.log').read().strip().splitlines()[-1]);print('G_MINW $v score_fund us', round(l['kernel_avg_us']['score_fund'],1), 'ms/step', round(l['ms_per_step'],2))"
  done
fi
