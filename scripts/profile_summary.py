#!/usr/bin/env python3
"""Per-step table from a rocprofv3 kernel trace of bench.py (scripts/gpu_profile.sh).
usage: profile_summary.py gpurun_out/prof_<tag>/bench_kernel_trace.csv [pairs]
A step starts at a knn2_partial_kernel launch over `pairs` problems (grid z); the window runs from the
fourth such launch (three warm-up steps) to the last one, so the PnP tail of every step inside it is seen."""
import collections, csv, re, sys

path = sys.argv[1]
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rows = list(csv.DictReader(open(path)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
starts = sorted(r["s"] for r in rows if "knn2_partial_kernel" in r["Kernel_Name"] and int(r["Grid_Size_Z"]) == pairs)
assert len(starts) >= 5, "expected three warm-up and at least two timed steps"
t0, t1, nsteps = starts[3], starts[-1], len(starts) - 4


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(<[^(]*?>)?)", name)
    return (m.group(1) if m else name)[:48]


acc, cnt = collections.Counter(), collections.Counter()
for r in rows:
    if t0 <= r["s"] < t1:
        k = short(r["Kernel_Name"])
        acc[k] += r["e"] - r["s"]
        cnt[k] += 1
print(f"wall span per step: {(t1 - t0) / nsteps / 1e6:.3f} ms ({nsteps} steps of {pairs} frame pairs, under rocprofv3 --kernel-trace)\n")
print("| kernel | launches/step | us/step (sum of launch durations) |\n|---|---|---|")
for k, v in acc.most_common():
    if v / nsteps / 1e3 < 5:
        continue
    print(f"| `{k}` | {cnt[k] / nsteps:g} | {v / nsteps / 1e3:.1f} |")
