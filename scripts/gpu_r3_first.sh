#!/bin/bash
# round 3, first GPU call: parity of the tree as it stands, kNN co-execution counters of the shipped kernel,
# F-scoring LDS counters, standalone solver rates, PnP-chain A/B.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out
B="python3 bench.py --inproc --no-cpu --no-latency"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3a_pytest.log 2>&1 || { tail -30 gpurun_out/r3a_pytest.log; exit 1; }
tail -3 gpurun_out/r3a_pytest.log
timeout -k 10 200 python3 scripts/diag/solver_rates.py gpurun_out/r3a_solver_rates.json > gpurun_out/r3a_solver_rates.log 2>&1 || { tail -20 gpurun_out/r3a_solver_rates.log; exit 1; }
cat gpurun_out/r3a_solver_rates.log
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 150 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/r3a_pmc_$tag -o pmc -- $B --steps 2 --warmup 1 > gpurun_out/r3a_pmc_$tag.log 2>&1 || { tail -5 gpurun_out/r3a_pmc_$tag.log; exit 1; }
  python3 - gpurun_out/r3a_pmc_$tag/pmc_counter_collection.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    s = "knn" if "knn2_partial" in k else "score_F" if "score_kernel<0>" in k else "p3p" if "solve_p3p" in k else "7pt" if "solve_7pt" in k else None
    if s:
        acc[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
for s in acc:
    print(s, {c: round(sum(v) / len(v)) for c, v in acc[s].items()}, "launches", len(next(iter(acc[s].values()))))
PY
done
$B --steps 10 --warmup 3 > gpurun_out/r3a_bench_tail.log 2>&1 && tail -1 gpurun_out/r3a_bench_tail.log | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('with pose tail', l['ms_per_step'], l['kernel_avg_us'])"
$B --steps 10 --warmup 3 --no-pose-tail > gpurun_out/r3a_bench_notail.log 2>&1 && tail -1 gpurun_out/r3a_bench_notail.log | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('no pose tail', l['ms_per_step'], l['kernel_avg_us'])"
$B --steps 10 --warmup 3 --no-pose-tail --no-pnp > gpurun_out/r3a_bench_nopnp.log 2>&1 && tail -1 gpurun_out/r3a_bench_nopnp.log | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('no pnp no tail', l['ms_per_step'], l['kernel_avg_us'])"
