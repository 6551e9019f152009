#!/bin/bash
# SQ counters of the F count kernel for several library builds (ablation): scripts/diag/pmc_libs.sh lib1.so lib2.so ...
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  VO_HIP_LIB=$lib timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmcab_$tag -o pmc -- python3 bench.py --inproc --no-cpu --no-latency --steps 2 --warmup 1 > gpurun_out/pmcab_$tag.log 2>&1 || { tail -5 gpurun_out/pmcab_$tag.log; exit 1; }
  python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/pmcab_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "score_kernel<0>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(tag, {k: f"{max(v):.4g}" for k, v in sorted(acc.items())})
PY
done
