#!/usr/bin/env python3
"""Timeline of one bench step from a rocprofv3 kernel trace: start offset, duration, queue, grid, kernel.
usage: step_timeline.py gpurun_out/prof_<tag>/bench_kernel_trace.csv [pairs] [step index]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
k = int(sys.argv[3]) if len(sys.argv) > 3 else 3
starts = sorted(int(r["Start_Timestamp"]) for r in rows if "knn2_partial_kernel" in r["Kernel_Name"] and int(r["Grid_Size_Z"]) == pairs)
t0, t1 = starts[k], starts[k + 1]
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 <= s < t1:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:56]
        print(f'{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} q{r["Queue_Id"]} g{r["Grid_Size_X"]}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]} {n}')
