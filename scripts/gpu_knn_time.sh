#!/bin/bash
# kernel-only timing of every kNN build (in-tree + ablate/libvo_knn_*.so); args: configs
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for lib in "" $(ls ablate/libvo_knn_*.so 2>/dev/null); do
  VO_HIP_LIB=${lib:+$PWD/$lib} timeout -k 10 200 python3 scripts/diag/knn_time.py ${@:-cfg4} 2>&1 | grep -v amdgpu.ids || exit 1
done
