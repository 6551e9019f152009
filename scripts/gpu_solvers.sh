#!/bin/bash
# solver parity tests + standalone solver rates.  usage: scripts/gpu_solvers.sh <tag>
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_ransac_gpu.py tests/test_golden_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/solvers_pytest.log 2>&1 || { tail -40 gpurun_out/solvers_pytest.log; exit 1; }
tail -2 gpurun_out/solvers_pytest.log
timeout -k 10 300 python3 scripts/diag/solver_rates.py gpurun_out/${1:-r3}_solver_rates.json 2>&1 | grep -v amdgpu.ids
