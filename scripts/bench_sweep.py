#!/usr/bin/env python3
"""Throughput of sweep.run_sweep INCLUDING the on-device generation of its frame pairs, one GPU.
usage: bench_sweep.py [pairs] [features per frame] [batch] [f_iters] [p_iters]   -> one JSON object
(profiles/r2_sweep.json holds the runs quoted in DESIGN.md)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visual_odometry_amd import _lib, sweep, synth  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
f_iters = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
p_iters = int(sys.argv[5]) if len(sys.argv) > 5 else 4096
_lib.require_device(0)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
K, size = synth.K_MALAGA, synth.IMAGE_SIZE["malaga"]
kw = dict(f_iters=f_iters, p_iters=p_iters, f_threshold=0.1, p_threshold=1.0, f_confidence=0.999, p_confidence=0.999)
sweep.run_sweep(list(range(10_000, 10_000 + batch)), n, n, batch, dev, **kw)          # warm-up (allocator, code objects)
torch.cuda.synchronize()
t0 = time.perf_counter()
tab = sweep.run_sweep(list(range(pairs)), n, n, batch, dev, **kw)
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
# generation alone, same batches
t0 = time.perf_counter()
for b0 in range(0, pairs, batch):
    sweep.frame_pairs_device(list(range(b0, min(pairs, b0 + batch))), n, n, K, size, dev)
torch.cuda.synchronize()
t_gen = time.perf_counter() - t0
print(json.dumps({"pairs": pairs, "features_per_frame": n, "batch": batch, "f_iters": f_iters, "p_iters": p_iters,
                  "seconds": t_all, "pairs_per_s_including_generation": pairs / t_all, "generation_seconds": t_gen,
                  "generation_share": t_gen / t_all, "median_rot_err_deg": float(sorted(tab["rot_err_deg"])[len(tab) // 2]),
                  "median_f_inliers": int(sorted(tab["f_inliers"])[len(tab) // 2]), "device": torch.cuda.get_device_name(0)}))
