#!/bin/bash
# Everything the bench's roofline objects cite, on the GPU box, from ONE command line of bench.py:
#   1. scripts/ubench/clock: in-kernel clock + cycles per VALU instruction (gpurun_out/r2_clock.txt)
#   2. rocprofv3 --kernel-trace --stats   (gpurun_out/prof_<tag>)
#   3. three separate --pmc passes: SQ instruction/cycle counters + GRBM_GUI_ACTIVE, FETCH_SIZE, WRITE_SIZE
# then scripts/make_counters.py folds 2 and 3 into profiles/r2_counters.json (copied back via gpurun_out/).
# usage: scripts/gpu_roofline.sh <tag> [bench.py args, e.g. --config cfg5]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=${1:-r2}; shift
mkdir -p gpurun_out
B="python3 bench.py --inproc --no-cpu --no-latency $*"
if [ -x scripts/ubench/clock ]; then timeout -k 10 120 scripts/ubench/clock > gpurun_out/${tag}_clock.txt 2>&1 || exit 1; fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o bench -- $B --steps 20 --warmup 3 > gpurun_out/prof_$tag.log 2>&1 || { tail -30 gpurun_out/prof_$tag.log; exit 1; }
grep "^{\"metric\"" gpurun_out/prof_$tag.log | tail -1 > gpurun_out/${tag}_bench_line.json
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${tag}_sq -o pmc -- $B --steps 2 --warmup 1 > gpurun_out/pmc_${tag}_sq.log 2>&1 || { tail -20 gpurun_out/pmc_${tag}_sq.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_f -o pmc -- $B --steps 2 --warmup 1 > gpurun_out/pmc_${tag}_f.log 2>&1 || { tail -20 gpurun_out/pmc_${tag}_f.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${tag}_w -o pmc -- $B --steps 2 --warmup 1 > gpurun_out/pmc_${tag}_w.log 2>&1 || { tail -20 gpurun_out/pmc_${tag}_w.log; exit 1; }
python3 scripts/make_counters.py $tag "$*" > gpurun_out/${tag}_counters.json && cat gpurun_out/${tag}_counters.json | head -60
