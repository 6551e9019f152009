#!/bin/bash
# Builds whole-library variants of the F count kernel for same-box A/B runs (scripts/gpu_ab_libs.sh "ablate/libvo_mf_*.so"):
#   usage: mf_variants.sh name:"-DFLAG=..." [name:"flags" ...]      -> ablate/libvo_mf_<name>.so
# The VO_SCORE_MF_ABLATE=1..4 timing branches (wrong counts) are not in the shipped header: apply
# scripts/experiments/mf_ablation_branches.patch (patch -p1) to a working copy first and revert afterwards.
cd "$(dirname "$0")/../visual_odometry_amd/csrc" || exit 1
mkdir -p ../../ablate/objs
OTHERS=$(ls _obj/*.o | grep -v vo_score.o)
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w -fno-slp-vectorize $flags \
     -c vo_score.hip -o ../../ablate/objs/vo_score_mf_$name.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../../ablate/objs/vo_score_mf_$name.o $OTHERS -o ../../ablate/libvo_mf_$name.so || exit 1
  echo built ablate/libvo_mf_$name.so
done
