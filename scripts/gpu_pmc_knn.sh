# SQ / L2 / HBM counters of the kNN kernels over a short bench run (three separate --pmc passes); used for the
# analysis recorded in scripts/experiments/vo_knn_segments.hip.txt and DESIGN.md 4.1
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
tag=$(echo $pass | cut -c1-6 | tr ' ' '_')
timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pmc_knn_$tag -o pmc -- python3 bench.py --inproc --steps 2 --warmup 1 --no-cpu --no-latency > gpurun_out/pmc_knn_$tag.log 2>&1 || { tail -5 gpurun_out/pmc_knn_$tag.log; exit 1; }
python3 - gpurun_out/pmc_knn_$tag/pmc_counter_collection.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    s = "sweep" if ("knn2_sweep" in k or "knn2_partial" in k) else "rescan" if "knn2_rescan" in k else None
    if s:
        acc[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[s]["dur_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for s in acc:
    print(s, {c: round(sum(v) / len(v)) for c, v in acc[s].items()})
PY
done
