#!/bin/bash
# Run on the GPU box (via gpurun): GPU parity tests, smoke, then a short bench.  Each GPU step has
# its own time limit and the script stops at the first failure.
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -25 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python -u __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
tail -2 gpurun_out/smoke.log
timeout -k 10 300 python -u bench.py --steps 5 --warmup 2 "$@" > gpurun_out/bench.log 2>&1 || { tail -20 gpurun_out/bench.log; exit 1; }
tail -3 gpurun_out/bench.log
